"""Where the time of a step goes on the GPU's timeline: reads a rocprofv3 --kernel-trace CSV (start / end timestamp of
every kernel) of a bench.py run, cuts it into steps at a marker kernel, and prints per step the span, the busy time (union
of the kernels' intervals), the idle time, and the largest gaps with the kernels on either side.

    rocprofv3 --kernel-trace -d out -o t --output-format csv -- python3 bench.py --steps 20 ... ;  python tools/step_gaps.py out [marker kernel] [first|last]
"""
import collections
import csv
import glob
import os
import sys


def main():
    files = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for r in csv.DictReader(open(files[0])):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        name = name.split("(")[0][:44]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Stream_Id", r.get("Queue_Id", "?"))))
    rows.sort()
    marker = sys.argv[2] if len(sys.argv) > 2 else "alignment_fwd_kernel"
    starts = [i for i, r in enumerate(rows) if r[2].startswith(marker)]
    steps = [(starts[i], starts[i + 1]) for i in range(len(starts) - 1)]
    # bench.py: warm-up, the timed (replayed) region, then a short eager region with stage timers: `part` picks where
    part = sys.argv[3] if len(sys.argv) > 3 else "first"
    steps = steps[len(steps) // 4:][:10] if part == "first" else steps[-12:-2]
    agg = collections.Counter()
    tot = collections.Counter()
    for a, b in steps:
        seg = rows[a:b]
        t0, t1 = seg[0][0], rows[b][0]
        busy, cur_s, cur_e, last_n = 0, seg[0][0], seg[0][1], seg[0][2]     # last_n: the kernel that ended last so far
        gaps = []
        for s, e, n, q in seg[1:]:
            if s > cur_e:
                busy += cur_e - cur_s
                gaps.append((s - cur_e, last_n, n))
                cur_s, cur_e, last_n = s, e, n
            elif e > cur_e:
                cur_e, last_n = e, n
        busy += cur_e - cur_s
        if t1 > cur_e:
            gaps.append((t1 - cur_e, last_n, "(next step)"))
        tot["span"] += t1 - t0
        tot["busy"] += busy
        tot["kernels"] += len(seg)
        for g, p, n in gaps:
            agg[(p, n)] += g
    n = max(len(steps), 1)
    print(f"{n} steps: span {tot['span'] / n / 1e3:.1f} us, busy {tot['busy'] / n / 1e3:.1f} us, idle "
          f"{(tot['span'] - tot['busy']) / n / 1e3:.1f} us, {tot['kernels'] / n:.0f} kernels per step")
    print("largest idle gaps per step (us): after kernel -> before kernel")
    for (p, nx), g in agg.most_common(25):
        print(f"  {g / n / 1e3:8.1f}   {p}  ->  {nx}")


if __name__ == "__main__":
    main()
