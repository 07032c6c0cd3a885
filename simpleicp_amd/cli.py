"""Command-line front end with the option names of the reference's C++ / Rust CLIs
(/root/reference/c++/src/simpleicp-cli.cpp:12-35, rust/src/main.rs:8-46) so that
scripts/benchmark.sh-style harnesses can drive the GPU build:

    python -m simpleicp_amd -f fixed.xyz -m movable.xyz [-c 1000 -n 10 -p 0.3 -o -1 -i 1 -x 100]

Output: the Python reference's log lines (iteration table, H, parameter table,
`Finished in N seconds!`), plus optional `--output` of the transformed movable cloud.
"""
from __future__ import annotations

import argparse
import sys

import numpy as np


def build_parser():
    ap = argparse.ArgumentParser(prog="simpleicp", description="A simple version of the ICP algorithm (MI355X build).")
    ap.add_argument("-f", "--fixed", required=True, help="Path to fixed point cloud (.xyz)")
    ap.add_argument("-m", "--movable", required=True, help="Path to movable point cloud (.xyz)")
    ap.add_argument("-c", "--correspondences", type=int, default=1000, help="Number of initially selected correspondences")
    ap.add_argument("-n", "--neighbors", type=int, default=10, help="Number of neighbors used for plane estimation")
    ap.add_argument("-p", "--min_planarity", type=float, default=0.3,
                    help="Minimal planarity value of planes used as correspondence")
    ap.add_argument("-o", "--max_overlap_distance", type=float, default=-1.0,
                    help="Maximum initial overlap distance. Set to negative value if point clouds are fully overlapping.")
    ap.add_argument("-i", "--min_change", type=float, default=1.0,
                    help="Minimal change of mean and standard deviation of distances (in percent) needed to proceed "
                         "to next iteration")
    ap.add_argument("-x", "--max_iterations", type=int, default=100, help="Maximum number of iterations")
    ap.add_argument("--output", default="", help="write the transformed movable cloud to this .xyz file")
    ap.add_argument("--quiet", action="store_true", help="print only the 4x4 matrix")
    return ap


def main(argv=None) -> int:
    args = build_parser().parse_args(argv)
    from . import PointCloud, SimpleICP, SimpleICPException, io
    from ._lib import BackendError
    try:
        X_fix = io.read_xyz(args.fixed)
        X_mov = io.read_xyz(args.movable)
        icp = SimpleICP(verbose=not args.quiet)
        icp.add_point_clouds(PointCloud(X_fix, columns=["x", "y", "z"]), PointCloud(X_mov, columns=["x", "y", "z"]))
        H, X_out, _, _ = icp.run(
            correspondences=args.correspondences, neighbors=args.neighbors, min_planarity=args.min_planarity,
            max_overlap_distance=args.max_overlap_distance if args.max_overlap_distance >= 0 else np.inf,
            min_change=args.min_change, max_iterations=args.max_iterations)
    except (SimpleICPException, BackendError, OSError) as exc:
        print(f"Caught exception: {exc}", file=sys.stderr)
        return 1
    if args.quiet:
        for row in H:
            print(" ".join(f"{v:.9f}" for v in row))
    if args.output:
        io.write_xyz(args.output, X_out)
    return 0


if __name__ == "__main__":
    sys.exit(main())
