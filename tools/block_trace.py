"""Per-block timeline of the pair-symmetric fp32 kernel (NB_SYM_TRACE): start / end tick of every block on the constant-rate counter and the
XCD / SE / CU it ran on. Answers: are the resident slots equally fast, are the blocks evenly placed?   python tools/block_trace.py n [n ...]"""
import os, sys, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
path = "/tmp/nb_sym_trace.txt"
os.environ["NB_SYM_TRACE"] = path
import bachelor_thesis_n_body_amd as nb

for n in [int(v) for v in sys.argv[1:]] or [65536]:
    if os.path.exists(path):
        os.remove(path)
    with nb.Engine(n, np.float32) as e:
        e.set_state(nb.plummer_sphere(n, 0, np.float32))
        e.accel(0.01, fetch=False)
        probe = nb.ClockProbe(0, 3000.0) if os.environ.get("BLOCK_TRACE_PROBE") else None      # the bench's clock probe beside the launch: one sleeping wave per XCD
        e.accel(0.01, fetch=False)
        if probe is not None:
            probe.stop(0)
            print("  clock probe beside the launch:", [round(v) for v in probe.end_xcd()[2]])
    launches, cur = [], None
    for line in open(path):
        if line.startswith("#"):
            cur = []; launches.append((line.strip(), cur))
        else:
            cur.append([int(v) for v in line.split()])
    head, rows = launches[-1]
    a = np.array(rows, dtype=np.int64)
    t0 = a[:, 1].min()
    start, end = (a[:, 1] - t0) / 100.0, (a[:, 2] - t0) / 100.0          # 100 MHz counter -> microseconds
    hw, xcc = a[:, 3], a[:, 4]
    cu, sh, se = (hw >> 8) & 0xf, (hw >> 12) & 0x1, (hw >> 13) & 0x7
    dur = end - start
    print(head)
    print(f"  makespan {end.max():.1f} us; block duration mean {dur.mean():.1f} min {dur.min():.1f} max {dur.max():.1f} us; start spread {start.max():.1f} us; "
          f"sum of durations / (makespan x blocks) = {dur.sum() / (end.max() * len(dur)):.3f}")
    per_xcc = collections.defaultdict(list)
    for i in range(len(dur)):
        per_xcc[int(xcc[i])].append(i)
    for x in sorted(per_xcc):
        idx = per_xcc[x]
        cus = collections.Counter((int(se[i]), int(sh[i]), int(cu[i])) for i in idx)
        print(f"  XCD {x}: {len(idx)} blocks on {len(cus)} CUs (blocks per CU: {sorted(collections.Counter(cus.values()).items())}); duration mean {dur[idx].mean():.1f} max {dur[idx].max():.1f}; last end {end[idx].max():.1f}")
    # which XCD gets block k: the dispatcher deals consecutive workgroups to consecutive XCDs — a fixed map k % 8 -> XCC_ID?
    res = collections.Counter((int(a[i, 0]) % 8, int(xcc[i])) for i in range(len(dur)))
    print("  block index mod 8 -> XCC_ID (count):", sorted(res.items()))
    # the XCDs finish apart: how much of the makespan is that (sum of a XCD's block durations = its busy time)
    busy = {x: dur[per_xcc[x]].sum() for x in per_xcc}
    mean_busy = sum(busy.values()) / len(busy)
    print("  busy time of each XCD relative to the mean:", {x: round(busy[x] / mean_busy, 4) for x in sorted(busy)},
          "; last end relative to the mean last end:", {x: round(end[per_xcc[x]].max() / np.mean([end[per_xcc[y]].max() for y in per_xcc]), 4) for x in sorted(per_xcc)})
    # blocks alone on their CU against blocks that shared it
    place = collections.Counter((int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i])) for i in range(len(dur)))
    alone = np.array([place[(int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i]))] == 1 for i in range(len(dur))])
    if alone.any() and (~alone).any():
        print(f"  blocks alone on a CU: {alone.sum()} (mean {dur[alone].mean():.1f} us); sharing a CU: {(~alone).sum()} (mean {dur[~alone].mean():.1f} us)")
    # the two blocks of a CU: wave slot of each (HW_ID[3:0] of the block's first wave) and their durations
    by_cu = collections.defaultdict(list)
    for i in range(len(dur)):
        by_cu[(int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i]))].append((float(start[i]), float(dur[i]), int(hw[i] & 0xf), int((hw[i] >> 4) & 0x3)))
    pairs = [sorted(v) for v in by_cu.values() if len(v) == 2 and abs(v[0][0] - v[1][0]) < 50.0]
    if pairs:
        d0, d1 = np.array([p[0][1] for p in pairs]), np.array([p[1][1] for p in pairs])
        print(f"  CUs whose two blocks started together: {len(pairs)}; shorter of the two mean {np.minimum(d0, d1).mean():.1f} us, longer mean {np.maximum(d0, d1).mean():.1f} us; "
              f"wave slots (first, second block): {sorted(collections.Counter((p[0][2], p[1][2]) for p in pairs).items())}; SIMD of wave 0: {sorted(collections.Counter((p[0][3], p[1][3]) for p in pairs).items())}")
        print("  first five pairs (start, duration, slot, simd):", pairs[:5])
    # the slowest CUs (sum of their blocks' durations): does a disturbance sit on a few CUs?
    per_cu = sorted(((sum(v[1] for v in vs), k) for k, vs in by_cu.items()), reverse=True)
    print("  CU busy time (sum of its blocks' durations) max / median / min: %.1f / %.1f / %.1f us; the 10 busiest CUs (xcd, se, sh, cu): %s" % (
        per_cu[0][0], per_cu[len(per_cu) // 2][0], per_cu[-1][0], [k for _, k in per_cu[:10]]))
    late = start > 0.2 * end.max()
    print(f"  blocks that started after 20 % of the makespan: {late.sum()}")
    order = np.argsort(end)
    print("  ten last blocks to finish (block, start, end, xcd, se, sh, cu):", [(int(a[i, 0]), round(float(start[i]), 1), round(float(end[i]), 1), int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i])) for i in order[-10:]])
