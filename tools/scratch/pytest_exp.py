"""Runs pytest with the -DGTAV_EXPERIMENTS library loaded first (debugging only: e.g. to see whether a kernel behaves the same in
both builds).  Usage: python tools/scratch/pytest_exp.py <pytest args>"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401  (the HIP runtime must come up through torch first)
import gtav_amd.lib as L
L.load_experiments()
import pytest
sys.exit(pytest.main(sys.argv[1:]))
