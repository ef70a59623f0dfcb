"""Per-kernel HBM roofline of one PM force evaluation (csrc/mesh.hip) from a rocprofv3 --kernel-trace --stats run of tools/bench_mesh.py.

Algorithmic bytes (what the step HAS to move through HBM; n particles, c = ngrid^3 cells, fp32 particles, fp64 complex grids of 16 B per
cell holding the two interlaced lattices):
  brick_key        12 n read (positions) + 4 n written (keys)                                    [+ 4 n index]
  radix sort       (4 + 4) n read + (4 + 4) n written, once                                      (rocPRIM: several passes — its own overhead)
  sort_gather      4 n (index) + 16 n gathered + 16 n written
  brick_deposit    16 n read + 16 c written — the deposit kernels AND brick_assemble_kernel (meshes of whole bricks: the tiles go through one
                   staging slot per chunk and the grid is written once, not cleared and not touched by atomics; other meshes: a fill clears
                   the grid, + 16 c, and the tiles are flushed with atomics)
  FFT (4 x 3-D)    a 3-D transform of a grid that does not fit in cache is three axis passes: 3 x (16 c read + 16 c written) each
  k-space          16 c read + 48 c written
  brick_gather     16 n read + 48 c read (three packed force grids) + 12 n written
Reports, per kernel group: calls per evaluation, microseconds per evaluation, GB/s achieved on the algorithmic bytes, fraction of the 8 TB/s
peak (6.3 TB/s is what a streaming kernel reaches on this chip: MI355X_MICROARCH.md)."""
import csv, json, sys

path, n_side, kind, evals = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
n, c = n_side ** 3, n_side ** 3
rows = list(csv.DictReader(open(path)))
groups = [
    ("brick_key", lambda s: "brick_key_kernel" in s, 12 * n + 8 * n),
    ("radix_sort (rocPRIM)", lambda s: "rocprim" in s.lower(), 16 * n),
    ("sort_gather", lambda s: "sort_gather_kernel" in s, 36 * n),
    ("grid clear (fill)", lambda s: "fillBuffer" in s, 16 * c),
    ("brick_deposit (+ assemble)", lambda s: "brick_deposit_kernel" in s or "mesh_deposit_kernel" in s or "brick_assemble_kernel" in s, 16 * n + 16 * c),
    ("fft (1 forward + 3 inverse, 3 axis passes each)", lambda s: s.startswith("fft_") or "transpose_rtc" in s, 4 * 3 * 32 * c),
    ("k-space", lambda s: "mesh_kspace_kernel" in s, 64 * c),
    ("brick_gather", lambda s: "brick_gather_kernel" in s or "mesh_gather_kernel" in s, 16 * n + 48 * c + 12 * n),
]
out, total_us, total_bytes = [], 0.0, 0
seen = set()
for name, match, nbytes in groups:
    sel = [r for r in rows if match(r["Name"])]
    if not sel:
        continue
    seen.update(r["Name"] for r in sel)
    us = sum(float(r["TotalDurationNs"]) for r in sel) / 1e3 / evals
    calls = sum(int(r["Calls"]) for r in sel) / evals
    out.append({"kernel": name, "launches_per_evaluation": round(calls, 1), "us_per_evaluation": round(us, 1), "algorithmic_bytes": nbytes,
                "achieved_GBps": round(nbytes / us / 1e3, 1), "frac_of_8TBps": round(nbytes / us / 1e3 / 8000.0, 4)})
    total_us += us; total_bytes += nbytes
other = sum(float(r["TotalDurationNs"]) for r in rows if r["Name"] not in seen and "cell_" not in r["Name"]) / 1e3 / evals
print(json.dumps({"config": f"PM force evaluation, {n_side}^3 particles ({kind}), {n_side}^3 mesh, fp32 particles / fp64 grids, {evals} evaluations profiled",
                  "kernels": out, "sum_us_per_evaluation": round(total_us, 1), "other_kernels_us": round(other, 1), "algorithmic_bytes_total": total_bytes,
                  "achieved_GBps_total": round(total_bytes / total_us / 1e3, 1), "frac_of_8TBps_total": round(total_bytes / total_us / 1e3 / 8000.0, 4)}))
