"""Writes tests/golden/synthetic256.json: the frozen definition of workload C5's scene
(256 spheres, SplitMix64 seed 0xC0FFEE).  The scene is the build's own
-- it does not exist in the reference -- so the fixture, not the generator, is the
authority once committed.  Floats are stored with repr() precision (round-trip exact).

    python tests/golden/make_synthetic.py
"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import workloads  # noqa: E402

if __name__ == "__main__":
    spheres = workloads.generate_synthetic(256, 0xC0FFEE)
    with open(workloads.SYNTH_FIXTURE, "w") as f:
        json.dump({"seed": "0xC0FFEE", "generator": "splitmix64", "spheres": spheres}, f, indent=0)
    print("wrote", workloads.SYNTH_FIXTURE, len(spheres))
