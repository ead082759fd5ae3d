"""Throughput of the BASELINE.json configurations other than the headline one (single GPU, reduced spp where the
full sample count would take minutes; samples are independent, so samples/s does not depend on spp).  The same runs are part
of bench.py's default line (`other_configs`); this script runs them alone.
usage: python scripts/bench_configs.py [C1 C3 C4 C5]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import bench
from cuda_path_tracer_amd import generators as G
from cuda_path_tracer_amd.scene import Scene, make_config

for name in sys.argv[1:] or ["C1", "C3", "C4", "C5"]:
    print(json.dumps(bench.run_other_config(name, G, Scene, make_config)), flush=True)
