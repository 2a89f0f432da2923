"""`python -m unconfined_amd deck.in [--mode faithful|fast] [--out FILE]`

What `./unconfined deck.in` does (reference driver.f90): read the 18-line deck and its time / space
file, compute drawdown and its log-time derivative for every requested point -- on the GPU --
and write the result file named on the deck's last line, in the reference's own format.
"""
import argparse
import sys

from . import output
from .engine import run_deck


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="python -m unconfined_amd", description=__doc__)
    ap.add_argument("deck", nargs="?", default="input.dat")
    ap.add_argument("--mode", default="fast", choices=["faithful", "fast"])
    ap.add_argument("--out", default=None, help="output file (default: the name on the deck's line 18)")
    args = ap.parse_args(argv)
    res = run_deck(args.deck, mode=args.mode)
    dk, D = res.deck, res.plan.derived
    if dk.timeseries:
        lines = output.timeseries_header(dk, D, res.r_dim, res.r_dim / D.Lc, res.z_dim[0], res.z_dim[0] / D.Lc, len(res.t))
        lines += output.timeseries_rows(res.t, res.h, res.dh)
    else:
        lines = output.contour_header(dk, D, res.r_dim_all, res.z_dim, res.t_dim, res.t_dim / D.Tc)
        lines += output.contour_rows(res.z, res.r, res.h[0], res.dh[0])
    output.write_lines(args.out or dk.outFileName, lines)
    return 0


if __name__ == "__main__":
    sys.exit(main())
