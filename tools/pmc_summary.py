#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc passes: tools/pmc_summary.py <counter_collection.csv> [more csv ...] [--json out.json]
Keeps the search-path kernels (k_trunk*, k_fc, k_step*, k_search).  FETCH_SIZE / WRITE_SIZE are in KiB per dispatch; on
gfx950 FETCH_SIZE counts half of a wide (16 B per lane) coalesced read stream (MI355X_MICROARCH.md), which the reader of the
numbers has to double -- this script reports the raw counter."""
import collections, csv, json, sys
args = [a for a in sys.argv[1:]]
out = None
if "--json" in args:
    i = args.index("--json"); out = args[i + 1]; del args[i:i + 2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for path in args:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not any(t in k for t in ("k_trunk", "k_fc", "k_step", "k_search", "k_split")):
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[k] = {"grid": int(r["Grid_Size"]), "workgroup": int(r["Workgroup_Size"]), "lds_bytes": int(r["LDS_Block_Size"]),
                   "vgprs": int(r["VGPR_Count"]), "agprs": int(r["Accum_VGPR_Count"])}
res = {}
for k, d in acc.items():
    res[k] = dict(meta[k])
    for c, v in d.items():
        res[k][c] = sum(v) / len(v)
        res[k]["dispatches"] = len(v)
    g = res[k]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in g and "GRBM_GUI_ACTIVE" in g and g["GRBM_GUI_ACTIVE"] > 0:
        # normalisation of profiles/r01_pmc_summary.json: the counter reads 128 x GRBM_GUI_ACTIVE when every MFMA pipe of the
        # chip is busy for the whole dispatch (1024 pipes, counted in units of 8 pipe-cycles)
        g["mfma_busy_fraction"] = g["SQ_VALU_MFMA_BUSY_CYCLES"] / (g["GRBM_GUI_ACTIVE"] * 128.0)
    if "SQ_LDS_BANK_CONFLICT" in g and g.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
        g["lds_conflict_fraction"] = g["SQ_LDS_BANK_CONFLICT"] / g["SQ_LDS_IDX_ACTIVE"]
    print(k, json.dumps({c: (round(v, 4) if isinstance(v, float) else v) for c, v in g.items()}))
if out:
    json.dump(res, open(out, "w"), indent=1)
