#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files per kernel (sum over dispatches, per-dispatch mean)."""
import csv
import glob
import json
import sys
from collections import defaultdict


def load(dirname):
    rows = []
    for p in glob.glob(dirname + "/**/*counter_collection.csv", recursive=True):
        with open(p) as f:
            rows += list(csv.DictReader(f))
    return rows


def short(name):
    n = name.split("(")[0]
    return n.replace("pt::", "")


def summarise(dirname):
    agg = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    for r in load(dirname):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    return {k: dict(v, dispatches=len(disp[k])) for k, v in agg.items()}


if __name__ == "__main__":
    out = {d: summarise(d) for d in sys.argv[1:]}
    print(json.dumps(out, indent=1))
