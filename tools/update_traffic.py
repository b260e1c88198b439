#!/usr/bin/env python3
"""Writes profiles/traffic.json: per bench entry the HBM traffic of its kernel
(rocprofv3 --pmc passes condensed by tools/summarize_pmc.py / summarize_profiles.py
into profiles/*_pmc.json) tied to the SHA-256 of the kernel's sources, so that
bench.py replays a ratio only for the version of the kernel it was taken on.
usage: update_traffic.py entry=pmc_json:needle[:bytes_key] ...
  needle = substring of the case line in the pmc json ("" = the file's single value)"""
import hashlib
import json
import os
import sys

REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
SOURCES = {
    "gemv": ["gemv_kernels.hpp"],
    "dot": ["dot_kernels.hpp"],
    "trsv": ["trsv_wide_kernels.hpp", "trsv_wide_detail.hpp", "trsv_wide_launch.hpp", "trsv_kernels.hpp"],
}


def sources_hash(names):
    digest = hashlib.sha256()
    for name in names:
        with open(os.path.join(REPO, "accessor-blas_amd", "csrc", name), "rb") as f:
            digest.update(f.read())
    return digest.hexdigest()


def main():
    path = os.path.join(REPO, "profiles", "traffic.json")
    table = {"kernels": {}}
    if os.path.exists(path):
        old = json.load(open(path))
        if "kernels" in old:
            table = old
    for arg in sys.argv[1:]:
        entry, _, rest = arg.partition("=")
        pmc, _, needle = rest.partition(":")
        summary = json.load(open(os.path.join(REPO, pmc)))
        if "cases" not in summary:
            # the headline's own summary (tools/summarize_profiles.py): one kernel
            summary = {"cases": [{
                "case": summary["kernel"],
                "traffic_over_algorithmic": summary["traffic_over_algorithmic"],
                "read_bytes_per_launch": summary["gemv_vec_bytes_per_launch"],
                "write_bytes_per_launch": 0.0}]}
        cases = summary["cases"]
        hits = [c for c in cases if needle in c["case"] and "traffic_over_algorithmic" in c]
        assert len(hits) == 1, (arg, [c["case"] for c in cases])
        case = hits[0]
        sources = SOURCES[entry.split("_")[0]]
        table["kernels"][entry] = {
            "source": pmc, "case": case["case"],
            "traffic_over_algorithmic": case["traffic_over_algorithmic"],
            "bytes_per_launch": case["read_bytes_per_launch"] + case["write_bytes_per_launch"],
            "sources": sources, "sources_sha256": sources_hash(sources)}
    json.dump(table, open(path, "w"), indent=1)
    print(json.dumps(table, indent=1))


if __name__ == "__main__":
    main()
