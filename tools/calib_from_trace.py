#!/usr/bin/env python3
"""Cross-check of bench.py's HIP-event calibration against rocprofv3: from a --kernel-trace CSV of the same bench command,
take the k_trunk dispatches that ran alone on the GPU (no other k_trunk dispatch of another queue overlapping them -- that is
the calibration ply, played by one engine while the others idle) and average their durations.
usage: calib_from_trace.py <kernel_trace.csv> <bench.json> <out.json>"""
import csv, json, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_trunk" in r["Kernel_Name"]]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]) for r in rows)
alone = []
for i, (s, e, q) in enumerate(ev):
    ok = True
    j = i - 1
    while j >= 0 and i - j < 16:
        if ev[j][1] > s and ev[j][2] != q: ok = False; break
        j -= 1
    j = i + 1
    while ok and j < len(ev) and ev[j][0] < e:
        if ev[j][2] != q: ok = False
        j += 1
    if ok: alone.append((e - s, q))
bench = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
by_q = {}
for d, q in alone: by_q.setdefault(q, []).append(d)
q_best = max(by_q, key=lambda q: len(by_q[q]))
durs = by_q[q_best]
out = {"source": "rocprofv3 --kernel-trace of `python3 bench.py --steps 2 --warmup 1 --no-episode --no-cpu` (default: 4 engines)",
       "kernel": rows[0]["Kernel_Name"].split("(")[0], "k_trunk_dispatches_total": len(ev),
       "dispatches_running_alone": len(durs), "avg_us_alone_rocprof": sum(durs) / len(durs) / 1e3,
       "avg_us_hip_events_bench": bench["roofline"]["avg_launch_ms"] * 1e3,
       "avg_us_all_dispatches_rocprof": sum(e - s for s, e, _ in ev) / len(ev) / 1e3,
       "note": "the dispatches that run alone are the calibration ply (one engine plays, the others idle); all other dispatches "
               "overlap with launches of the other three engines, which stretches their individual durations"}
out["ratio_rocprof_over_events"] = out["avg_us_alone_rocprof"] / out["avg_us_hip_events_bench"]
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
