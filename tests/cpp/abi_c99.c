/* tests/cpp/abi_c99.c — include/glprover.h must be a plain C header (no C++ types): this file is
 * compiled with `gcc -std=c99 -pedantic -Wall -Werror` and linked against libglprover.so by
 * tests/test_abi.py; it only takes addresses, it does not need a GPU. */
#include <stdio.h>
#include "glprover.h"

int main(void) {
    glp_ctx* ctx = 0;
    glp_fri_config cfg;
    glp_fri_batch b;
    cfg.log_n = 0; b.n_polys = 0;
    (void)cfg; (void)b;
    /* without a device this must fail cleanly, never abort */
    int rc = glp_create(&ctx, 0);
    printf("%s rc=%d ctx=%s\n", glp_version(), rc, ctx ? "set" : "null");
    if (rc == GLP_OK) glp_destroy(ctx);
    return (rc == GLP_OK || rc == GLP_E_NODEVICE || rc == GLP_E_HIP) ? 0 : 1;
}
