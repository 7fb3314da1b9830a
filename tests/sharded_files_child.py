"""Child process of tests/test_sharded_files.py: tools/run_sharded_pipeline.py (BASELINE config C4's driver) with the
oracle standing in for the GPU leg, so that its sharding, resume and counter logic runs on CPU under torchrun / gloo.
Test infrastructure: the shipped driver has no such switch."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))

import run_sharded_pipeline  # noqa: E402
from test_pipeline import _oracle_process  # noqa: E402

if __name__ == "__main__":
    check_cols = 4 if "kitti" in sys.argv else 5
    run_sharded_pipeline.main(process=_oracle_process(check_cols))
