"""Time cw_run / cw_check_r1cs on prebuilt schedules: python tools/tape_bench.py <dir> <name> <batch> [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from circom_amd import runtime as rt
if os.environ.get("CW_LIB"):
    from pathlib import Path
    rt.LIB_PATH = Path(os.environ["CW_LIB"]).resolve()
d, name, B = sys.argv[1], sys.argv[2], int(sys.argv[3])
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
c = rt.Circuit(os.path.join(d, name + ".cwt"), os.path.join(d, name + ".dat"), os.path.join(d, name + ".r1cs"))
rng = np.random.default_rng(1)
if name.startswith("sha"):
    arr = np.zeros((B, c.n_inputs, 32), dtype=np.uint8); arr[:, :, 0] = rng.integers(0, 2, size=(B, c.n_inputs), dtype=np.uint8)
else:
    arr = rng.integers(0, 256, size=(B, c.n_inputs, 32), dtype=np.uint8); arr[:, :, 31] &= 0x0F
stream = torch.cuda.current_stream()
b = c.batch(B, device=0, stream=stream.cuda_stream)
din = torch.from_numpy(arr).cuda()
b.set_inputs_device(din.data_ptr())
b.run(); b.check_r1cs(); torch.cuda.synchronize()
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]
for e in ev:
    e[0].record(stream); b.run(); e[1].record(stream); b.check_r1cs(); e[2].record(stream)
torch.cuda.synchronize()
print("TB %s %s B=%d S=%d pipe=%s L=%d eval %.3f ms r1cs %.3f ms" % (d, name, B, b.strands, b.pipelined, b.lanes, sum(e[0].elapsed_time(e[1]) for e in ev) / steps, sum(e[1].elapsed_time(e[2]) for e in ev) / steps))

if os.environ.get("CW_LIB"):
    import ctypes
    from circom_amd.hip_elements.lower import D_NAMES
    buf = (ctypes.c_ulonglong * (16 * 64 * 2))()
    L = rt.lib()
    if hasattr(L, "cw_debug_profile") and L.cw_debug_profile(buf, 0) == 0:
        tot = {}
        for w in range(16):
            line = []
            for op in range(64):
                t, n = buf[(w * 64 + op) * 2], buf[(w * 64 + op) * 2 + 1]
                if n:
                    nm = D_NAMES[op] if op < len(D_NAMES) else str(op)
                    line.append((t, nm, n))
                    a = tot.setdefault(nm, [0, 0]); a[0] += t; a[1] += n
            if line:
                s_ = sum(x[0] for x in line)
                print("PROF strand %2d total %.2f Mclk: " % (w, s_ / 1e6) + "  ".join("%s %.0f%% (%d x %.0f)" % (nm, 100.0 * t / s_, n, t / n) for t, nm, n in sorted(line, reverse=True)[:7]))
        s_ = sum(v[0] for v in tot.values())
        print("PROF all strands: " + "  ".join("%s %.1f%% (avg %.0f clk)" % (k, 100.0 * v[0] / s_, v[0] / v[1]) for k, v in sorted(tot.items(), key=lambda kv: -kv[1][0])))

    seg = (ctypes.c_ulonglong * (64 * 4))()
    if hasattr(L, "cw_debug_profile_seg") and L.cw_debug_profile_seg(seg) == 0:
        for op in range(64):
            n = seg[op * 4 + 3]
            if n:
                nm = D_NAMES[op] if op < len(D_NAMES) else str(op)
                print("SEG strand 0 %-10s x %6d: operands ready %6.0f clk | arithmetic %6.0f | destinations %6.0f" % (nm, n, seg[op * 4] / n, seg[op * 4 + 1] / n, seg[op * 4 + 2] / n))

    arr = (ctypes.c_ulonglong * (4096 * 16))()
    if hasattr(L, "cw_debug_profile_arrive") and L.cw_debug_profile_arrive(arr) == 0:
        import collections
        a = np.frombuffer(arr, dtype=np.uint64).reshape(4096, 16).astype(np.int64)
        S = b.strands
        lv = [k for k in range(4096) if (a[k, :S] > 0).all()]
        if len(lv) > 2:
            last = collections.Counter()
            spread, dur = [], []
            for k in lv:
                row = a[k, :S]
                last[int(row.argmax())] += 1
                spread.append(int(row.max() - row.min()))
            for k0, k1 in zip(lv[:-1], lv[1:]):
                if k1 == k0 + 1:
                    dur.append(int(a[k1, :S].max() - a[k0, :S].max()))
            print("ARRIVE %d levels: level duration avg %.0f clk (median %.0f); first-to-last arrival spread avg %.0f clk (median %.0f)" % (
                len(lv), np.mean(dur), np.median(dur), np.mean(spread), np.median(spread)))
            print("ARRIVE last strand at the barrier: " + "  ".join("s%d %.0f%%" % (s_, 100.0 * n / len(lv)) for s_, n in last.most_common(8)))
