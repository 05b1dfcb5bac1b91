#!/usr/bin/env python
# -*- coding: utf-8 -*-
"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel name.
    python tools/pmc_summary.py <dir> [<dir> ...]  -> prints/returns {kernel: {counter: mean per launch}}"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def summarize(dirs):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    name = row.get("Kernel_Name") or row.get("Kernel Name")
                    cnt = row.get("Counter_Name") or row.get("Counter Name")
                    val = float(row.get("Counter_Value") or row.get("Counter Value") or 0)
                    a = acc[name][cnt]
                    a[0] += val
                    a[1] += 1
    return {k: {c: v[0] / max(v[1], 1) for c, v in cs.items()} | {"launches": max(v[1] for v in cs.values())}
            for k, cs in acc.items()}


if __name__ == "__main__":
    out = summarize(sys.argv[1:])
    print(json.dumps(out, indent=1, sort_keys=True))
