"""profiles/rNN_pmc_summary.json from the three raw PMC passes (tools/rocpd_pmc.py output of separate rocprofv3 --pmc runs:
FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE).  Corrections as the guide's HBM section says and as
rounds 2-4 applied them: FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled for coalesced streaming reads on gfx950.
usage: python tools/pmc_summarize.py profiles/r05_pmc  ->  reads <prefix>_FETCH_SIZE.txt, _WRITE_SIZE.txt, _SQ_GRBM.txt, writes <prefix>_summary.json"""
import json, re, sys

def parse(path):
    out, cur = {}, None
    try:
        for ln in open(path):
            if not ln.startswith(" "):
                m = re.match(r"(.*?)\s+\(dispatches (\d+)\)", ln.strip())
                cur = m.group(1) if m else None
                if cur:
                    out[cur] = {"dispatches": int(m.group(2))}
            elif cur:
                p = ln.split()
                out[cur][p[0]] = float(p[1])
    except FileNotFoundError:
        pass
    return out

prefix = sys.argv[1]
F, W, S = parse(prefix + "_FETCH_SIZE.txt"), parse(prefix + "_WRITE_SIZE.txt"), parse(prefix + "_SQ_GRBM.txt")
ALG = {"gemm_f64_kernel<true, 4, 1>": 904020000}     # config 2: q n 8 + n (n + 1) / 2 * 8 bytes per Schur syrk launch
res = {"_about": "rocprofv3 --kernel-trace --pmc passes (ONE counter group per pass, kernel filter), summarised by tools/pmc_summarize.py; "
                 "FETCH_SIZE in KB doubled for coalesced streaming reads on gfx950, WRITE_SIZE in KB as is; mfma_pipe_busy_fraction = "
                 "SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs... as counted: / (GRBM_GUI_ACTIVE x 1024 / 8)) following rounds 2-4"}
for k in sorted(set(F) | set(W) | set(S)):
    e = {}
    if k in F and "FETCH_SIZE" in F[k]:
        e["FETCH_SIZE_KB_per_dispatch"] = F[k]["FETCH_SIZE"]
        e["fetch_bytes_corrected"] = F[k]["FETCH_SIZE"] * 1000.0 * 2.0
    if k in W and "WRITE_SIZE" in W[k]:
        e["WRITE_SIZE_KB_per_dispatch"] = W[k]["WRITE_SIZE"]
        e["write_bytes"] = W[k]["WRITE_SIZE"] * 1000.0
    if "fetch_bytes_corrected" in e and "write_bytes" in e:
        e["hbm_bytes_per_dispatch"] = e["fetch_bytes_corrected"] + e["write_bytes"]
    if k in S and S[k].get("GRBM_GUI_ACTIVE"):
        g, mf = S[k]["GRBM_GUI_ACTIVE"], S[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        e["GRBM_GUI_ACTIVE"] = g
        e["SQ_VALU_MFMA_BUSY_CYCLES"] = mf
        e["mfma_pipe_busy_fraction"] = mf / (g / 8.0 * 1024.0)     # (GUI_ACTIVE summed over 8 XCDs; 1024 SIMDs)
    for name, ab in ALG.items():
        if name in k:
            e["algorithmic_bytes_per_launch"] = ab
    e["dispatches"] = max(d.get(k, {}).get("dispatches", 0) for d in (F, W, S))
    res[k[:110]] = e
syrk = next((v for k, v in res.items() if "gemm_f64_kernel<true, 4, 1>" in k), None)
if syrk and "hbm_bytes_per_dispatch" in syrk:
    res["syrk"] = {"hbm_bytes_per_launch": syrk["hbm_bytes_per_dispatch"], "algorithmic_bytes_per_launch": 904020000,
                   "mfma_pipe_busy_fraction": syrk.get("mfma_pipe_busy_fraction")}
json.dump(res, open(prefix + "_summary.json", "w"), indent=1)
print(json.dumps(res.get("syrk"), indent=1))
