#!/usr/bin/env python3
"""profiles/isa_counts.py <file.s> — static instruction counts per kernel of a gfx950 assembly listing (hipcc --cuda-device-only -S):
VALU total, v_mov_b32, v_mad_u64_u32, SALU, s_nop, s_waitcnt (and how many are vmcnt(0)), global loads/stores, LDS reads/writes.
Measurement aid (the VERDICT r1 'ISA listing' evidence); kernels = text between a _Z...: label and its s_endpgm."""
import re
import sys

for path in sys.argv[1:]:
    lines = open(path).read().splitlines()
    name, buf = None, []
    for ln in lines:
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            name, buf = m.group(1), []
            continue
        if name is None:
            continue
        t = ln.split(";")[0].strip()
        if t:
            buf.append(t)
        if t.startswith("s_endpgm"):
            op = [b.split()[0] for b in buf]
            c = lambda pred: sum(1 for o in op if pred(o))
            vm0 = sum(1 for b in buf if b.startswith("s_waitcnt") and "vmcnt(0)" in b)
            print(f"{path}: {name[:60]:60s} VALU={c(lambda o: o.startswith('v_')):5d} v_mov_b32={c(lambda o: o.startswith('v_mov_b32')):4d} "
                  f"v_mad_u64_u32={c(lambda o: o == 'v_mad_u64_u32'):4d} SALU={c(lambda o: o.startswith('s_')):5d} s_nop={c(lambda o: o == 's_nop'):4d} "
                  f"s_waitcnt={c(lambda o: o == 's_waitcnt'):3d} (vmcnt(0): {vm0}) global_load={c(lambda o: o.startswith('global_load')):3d} "
                  f"global_store={c(lambda o: o.startswith('global_store')):3d} ds_read={c(lambda o: o.startswith('ds_read')):3d} ds_write={c(lambda o: o.startswith('ds_write')):3d}")
            name = None
