#!/usr/bin/env python
"""Build profiles/rNN_pmc_traffic.json: HBM bytes per launch per kernel family from two rocprofv3 --pmc passes
(FETCH_SIZE and WRITE_SIZE in separate runs, MI355X_MICROARCH.md §HBM: FETCH_SIZE is reported in KiB and, on gfx950,
counts wide coalesced reads at half their size -> x2; WRITE_SIZE in KiB, uncalibrated)."""
import json
import sqlite3
import sys

FAMILIES = {"linear_mfma_*": "linear_mfma", "gt_attn_fused_edge_fwd_kernel": "gt_attn_fused_edge", "layernorm_fwd_kernel": "layernorm_fwd",
            "edge_ln_res_segsum_kernel": "edge_ln_res_segsum", "gt_chain2_kernel": "gt_chain2_kernel", "gt_cluster_chain_kernel": "gt_cluster_chain_kernel", "gt_rowchain_kernel": "gt_rowchain_kernel", "gnn_edge_chain_kernel": "gnn_edge_chain_kernel",
            "gnn_node_chain_kernel": "gnn_node_chain_kernel", "segment_sum_rows_kernel": "segment_sum_rows_kernel"}


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    ix = {n: i for i, n in enumerate(cols)}
    name_col = "kernel_name" if "kernel_name" in ix else [n for n in cols if "kernel" in n and "name" in n][0]
    out = {}
    for r in c.execute("select * from counters_collection"):
        if r[ix["counter_name"]] != counter:
            continue
        for fam, pat in FAMILIES.items():
            if pat in r[ix[name_col]]:
                a = out.setdefault(fam, [0, 0.0])
                a[0] += 1
                a[1] += r[ix["value"]]
    return {k: v[1] / v[0] for k, v in out.items()}, {k: v[0] for k, v in out.items()}


fetch, n = per_kernel(sys.argv[1], "FETCH_SIZE")
write, _ = per_kernel(sys.argv[2], "WRITE_SIZE")
res = {}
detail = {}
for fam in FAMILIES:
    if fam in fetch and fam in write:
        rd, wr = fetch[fam] * 1024 * 2, write[fam] * 1024
        res[fam] = round(rd + wr)
        detail[fam] = {"launches_sampled": n[fam], "fetch_bytes_x2_corrected": round(rd), "write_bytes": round(wr)}
json.dump(res, open(sys.argv[3], "w"), indent=1)
json.dump(detail, open(sys.argv[3].replace(".json", "_detail.json"), "w"), indent=1)
print(json.dumps(detail, indent=1))
