#!/usr/bin/env python3
"""Design probes on the GPU with the defaults for large, highly diverse input:
python -m catch_amd.design_large a.fasta [...] -o probes.fasta

The wrapper of bin/design_large.py: catch_amd.design with the "large" profile
(-m 5, -e 50, --cluster-and-design-separately 0.15, --cluster-from-fragments
50000, --filter-with-lsh-minhash 0.6; bin/design.py:502, :583, :753, :794,
:846).  Every value can still be overridden on the command line.
"""
import sys

from catch_amd import design

if __name__ == "__main__":
    design.main(design.parse_args(sys.argv[1:], args_type="large"))
