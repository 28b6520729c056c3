#!/usr/bin/env python3
"""profiles/rN/traffic.json from the two PMC summaries: per contraction family, HBM-side bytes per launch =
(2 x FETCH_SIZE + WRITE_SIZE) KiB summed over the family's kernels / its launches (FETCH_SIZE doubled per the gfx950
correction of MI355X_MICROARCH.md, WRITE_SIZE as reported).
    python tools/make_traffic.py profiles/r1"""
import json
import re
import sys

FAM = {"0": "gemm", "1": "conv3x3", "2": "tconv3"}


def table(path, counter):
    out = {}
    for line in open(path):
        # kernel names: gemm_kernel<Geo<4, 2>, MODE, FAST, SB> (round 3 on; the tile geometry comes first) or gemm_kernel<MODE, ...>
        # (round 5: wgemm_kernel<MODE, NREP, GEGLU> — the 288 x 320 tile — and its persistent plain-GEMM form wgemm_pkernel<NREP, GEGLU>)
        m = re.match(r"\| `([pw]?gemm\w*_p?kernel)<(?:\(anonymous namespace\)::)?(?:Geo<\d+, \d+>, )?(\d)[^`]*` \| " + counter + r" \| (\d+) \| ([0-9.e+]+) \|", line)
        h = re.match(r"\| `(?:\(anonymous namespace\)::)?(hgeglu_kernel)<[^`]*` \| " + counter + r" \| (\d+) \| ([0-9.e+]+) \|", line)
        if h:           # round 6: the two-workgroup 144 x 256 GEGLU kernel (plain GEMM family)
            e = out.setdefault("gemm", {"kernels": [], "launches": 0, "kib": 0.0})
            e["kernels"].append("hgeglu_kernel<..>")
            e["launches"] += int(h.group(2))
            e["kib"] += float(h.group(3))
            continue
        if m:
            fam = "gemm" if m.group(1) == "wgemm_pkernel" else FAM[m.group(2)]
            e = out.setdefault(fam, {"kernels": [], "launches": 0, "kib": 0.0})
            e["kernels"].append(f"{m.group(1)}<{m.group(2)},..>")
            e["launches"] += int(m.group(3))
            e["kib"] += float(m.group(4))
    return out


def main():
    d = sys.argv[1].rstrip("/")
    fetch, write = table(f"{d}/pmc_fetch.md", "FETCH_SIZE"), table(f"{d}/pmc_write.md", "WRITE_SIZE")
    rec = {"round": d.split("/")[-1],
           "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of `python bench.py --steps 1 "
                     "--warmup 0 --no-graph --no-cpu-baseline --no-profile --no-decode`; per launch = (2 x FETCH_SIZE + WRITE_SIZE) "
                     "KiB summed over the family's kernels / launches (gfx950: FETCH_SIZE counts 64 B per 128-B request, "
                     "MI355X_MICROARCH.md HBM section; WRITE_SIZE as reported)",
           "families": {}}
    for fam, f in fetch.items():
        w = write.get(fam, {"kib": 0.0, "launches": f["launches"]})
        n = f["launches"]
        rec["families"][fam] = {"kernels": sorted(set(f["kernels"])), "launches": n,
                                # the traced run is ONE step (plus its one-off cross-attention K / V projections, which are
                                # small): bench.py divides this by the live launches per step
                                "bytes_per_step": round((2 * f["kib"] + w["kib"]) * 1024),
                                "fetch_bytes_per_launch": round(2 * f["kib"] * 1024 / n),
                                "write_bytes_per_launch": round(w["kib"] * 1024 / n),
                                "bytes_per_launch": round((2 * f["kib"] + w["kib"]) * 1024 / n)}
    json.dump(rec, open(f"{d}/traffic.json", "w"), indent=1)
    print(json.dumps(rec["families"], indent=1))


if __name__ == "__main__":
    main()
