#!/usr/bin/env python3
"""Regenerates the committed fixtures under tests/golden/ (run in the build
container only, where /root/reference exists).

Sources of truth:
  * structure fixtures (*.struct.json): output of oracle/_ref/ref_dump, i.e. the
    reference's OWN alignment / site-pattern / parser / node-id / Detrifurcate /
    traversal code run on the reference's own data files (copies of which sit in
    tests/golden/data/ because /root/reference does not exist on the GPU box).
  * reference_kats.json: the known-answer values written in the reference's
    tests (file:line cited per entry) -- data, typed in verbatim.
"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF_DUMP = os.path.join(REPO, "oracle", "_ref", "ref_dump")
DATA = os.path.join(HERE, "data")

CASES = {
    # name: (fasta, tree file, format, kind)
    "hello": ("hello.fasta", "hello.nwk", "newick", "unrooted"),
    "hello_out": ("hello.fasta", "hello_out.t", "nexus", "unrooted"),
    "ds1_sub10": ("DS1.fasta", "DS1.subsampled_10.t", "nexus", "unrooted"),
    "ds1_top100": ("DS1.fasta", "DS1.100_topologies.nwk", "newick", "unrooted"),
    "five_taxon": ("five_taxon.fasta", "five_taxon_unrooted.nwk", "newick", "unrooted"),
    "flua": ("fluA.fa", "fluA.tree", "newick", "rooted"),
}


def main():
    if not os.path.exists(REF_DUMP):
        sys.exit("build oracle/_ref/ref_dump first (make -C oracle ref)")
    for name, (fasta, tree, fmt, kind) in CASES.items():
        out = subprocess.run(
            [REF_DUMP, os.path.join(DATA, fasta), os.path.join(DATA, tree), fmt, kind],
            check=True, capture_output=True, text=True).stdout
        obj = json.loads(out)
        obj["source"] = {"fasta": fasta, "trees": tree, "format": fmt, "kind": kind,
                         "generator": "oracle/_ref/ref_dump (reference sources, unmodified)"}
        with open(os.path.join(HERE, name + ".struct.json"), "w") as fh:
            json.dump(obj, fh, separators=(",", ":"))
        print(name, obj["taxon_count"], obj["site_count"], obj["pattern_count"],
              len(obj["trees"]))


if __name__ == "__main__":
    main()
