import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import __graft_entry__ as graft
from bench import splitmix_fill
pkg = graft.load_package()
pr = pkg.Prover(0)
log_n, batch = 23, 32
d = pr.to_device(splitmix_fill(batch << log_n, 1).reshape(batch, 1 << log_n))
for plan in [None, "8:4,8:4,7:5", "8:4,7:5,8:4", "7:5,8:4,8:4", "12:2,11:3", "11:3,12:2", "10:4,7:5,6:6", "6:6,7:5,10:3", "10:3,7:5,6:6", "8:4,8:4,7:4", "12:2,6:4,5:6"]:
    try:
        pr.set_plan(log_n, plan)
    except Exception as e:
        print(plan, "ERR", e); continue
    for _ in range(2): pr.ntt_ex(d, d, log_n, batch, flags=pkg.NTT_BITREV)
    pr.sync(); pr.timer_start()
    for _ in range(5): pr.ntt_ex(d, d, log_n, batch, flags=pkg.NTT_BITREV)
    ms = pr.timer_stop() / 5
    pr.set_profiling(True); pr.ntt_ex(d, d, log_n, batch, flags=pkg.NTT_BITREV); pm = pr.last_pass_ms(); pr.set_profiling(False)
    print(json.dumps({"plan": pr.describe_plan(log_n, batch, pkg.NTT_BITREV), "ms": round(ms,3), "gbps": round(16.0*(1<<log_n)*batch/ms/1e6,1), "pass_ms": [round(x,3) for x in pm]}), flush=True)
pr.close()
