#!/usr/bin/env python3
"""profiles/summarize_timeline.py <kernel_trace.csv> [<memory_copy_trace.csv>] -- the device timeline of ONE single-frame call out of a
`rocprofv3 --kernel-trace [--memory-copy-trace]` run of tools/single_frame_probe.py: the dispatches of the call are found as the
last group whose first kernel matches --first (default k_resize2) and whose last matches --last (default k_describe_tiles_rare);
printed per dispatch: start offset from the group's first start, duration, gap to the previous end (idle GPU), all in microseconds,
and the median over all such groups of the run."""
import argparse
import csv
import statistics as st


def short(name):
    return name.split("(")[0].replace("void ", "").split("<")[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--first", default="k_resize2")
    ap.add_argument("--last", default="k_describe_tiles_rare")
    ap.add_argument("--max-span-us", type=float, default=2000.0)
    args = ap.parse_args()
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(args.trace))]
    rows.sort()
    groups, cur = [], None
    for i, (s, e, n) in enumerate(rows):
        if n == args.first and (cur is None or rows[i - 1][2] != args.first):
            cur = []
        if cur is not None:
            cur.append((s, e, n))
            if n == args.last:
                if (cur[-1][1] - cur[0][0]) / 1e3 < args.max_span_us:
                    groups.append(cur)
                cur = None
    if not groups:
        raise SystemExit("no dispatch group %s .. %s found" % (args.first, args.last))
    shape = [n for _, _, n in groups[-1]]
    same = [g for g in groups if [n for _, _, n in g] == shape]
    print("%d groups of %d dispatches with the shape of the last one (%s .. %s)" % (len(same), len(shape), args.first, args.last))
    print("%-26s %10s %10s %10s" % ("kernel", "start_us", "dur_us", "gap_us"))
    tot_d = tot_g = 0.0
    for j, n in enumerate(shape):
        so = st.median((g[j][0] - g[0][0]) / 1e3 for g in same)
        du = st.median((g[j][1] - g[j][0]) / 1e3 for g in same)
        ga = st.median(((g[j][0] - g[j - 1][1]) / 1e3 if j else 0.0) for g in same)
        tot_d += du; tot_g += ga
        print("%-26s %10.2f %10.2f %10.2f" % (n, so, du, ga))
    span = st.median((g[-1][1] - g[0][0]) / 1e3 for g in same)
    print("span %.2f us = kernels %.2f + gaps %.2f (medians per row; the span is the median of the groups' own spans)" % (span, tot_d, tot_g))


if __name__ == "__main__":
    main()
