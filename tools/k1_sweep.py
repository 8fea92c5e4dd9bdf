#!/usr/bin/env python3
"""K1 of SURVEY 8-d through the product path: one gate per launch on a 2 GiB state, every target
wire (bench.k1_sweep), printed per gate kind."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

for g, v in bench.k1_sweep().items():
    print(g, "bytes/amp", v["bytes_per_amplitude"], "ms min/mean/max", v["ms_min_mean_max"],
          "frac of 8 TB/s min/mean/max", v["frac_of_8TBps_min_mean_max"], "slowest wire", v["slowest_target_wire"])
    print("   ms per target wire 0..27:", v["ms_per_target_wire"])
    print("   vs attainable bytes: frac min/mean/max", v["frac_of_8TBps_vs_attainable_min_mean_max"],
          "wires at >= 0.70:", v["target_wires_at_0.70_or_more_of_attainable"], "of", len(v["ms_per_target_wire"]))
