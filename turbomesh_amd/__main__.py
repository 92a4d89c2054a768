"""`python -m turbomesh_amd <config.json>` -- the reference program's batch path (src/gui/main.zig:27-56 without the GUI):
parse the input file (input.zig:25-41), build the geometry, run the blocking template (TFI of every block on the GPU),
smooth, write the mesh.

The example inputs name the reference's own solvers ("gmres" + "ilu0"), which stay on the Zig side: like a reference build
without UMFPACK answers error.ExternalSolverNotEnabled, this program refuses them -- unless --hip replaces the solver entry
by {"hip": {"inner": "auto"}} (what a user would write into the JSON): the plain Picard + BiCGStab solve on meshes of small blocks
like the reference's examples (T106 / LS89: 7x faster there than the multigrid-preconditioned one), the multigrid-preconditioned
solve once a block has 100 000 nodes or more (1000 when the blocks are not coupled) and the cells' aspect ratio does not vary strongly inside any block (refined O-grids with
boundary-layer clustering keep the plain solve)."""
from __future__ import annotations

import argparse
import logging
import os
import sys

from . import input as tm_input
from .smoothing import smooth, solver


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m turbomesh_amd", description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("config", help="input file in the reference's JSON schema (examples/T106/T106.json)")
    ap.add_argument("--hip", nargs="?", const="auto", choices=["auto", "bicgstab", "mg_bicgstab", "gmres", "relax", "file"],
                    help="use the hip solver with this inner strategy instead of the solver named in the file; `file` = the device counterpart of "
                         "the solver the file names (gmres -> GMRES(30) on the device, bicgstab -> BiCGStab; ilu0 -> diagonal)")
    ap.add_argument("--iterations", type=int, help="override smoothing.iterations")
    ap.add_argument("--output", help="override the output file (.xyz / .p3d: multi-block PLOT3D)")
    ap.add_argument("--until", type=float, metavar="TOL",
                    help="iterate until the scaled nonlinear residual is <= TOL (at most `iterations`, default 100) instead of a fixed count")
    args = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO, format="%(levelname)s(%(name)s): %(message)s")

    with open(args.config) as f:
        inp = tm_input.Input.parse(f.read())
    if args.hip == "file":
        inp.solver, note = inp.solver.served_by_hip()
        if note:
            logging.getLogger("smoothing").warning(note)
    elif args.hip:
        inp.solver = solver.Option.hip(inner=getattr(solver.Inner, args.hip))
    if inp.solver.tag != solver.Tag.hip:
        sys.exit(f"error.ExternalSolverNotEnabled: solver `{inp.solver.tag.name}` is served by the Zig program; "
                 "use \"solver\": {\"hip\": {}} in the input file or pass --hip")
    geometry = inp.geometry(os.getcwd())   # profile files are named relative to the working directory, as in the reference
    mesh = inp.template.run(geometry)                                   # blocking (O4H.zig:67-118) + TFI per block
    iterations = inp.iterations if args.iterations is None else args.iterations
    slog = logging.getLogger("smoothing")
    with smooth.per_iteration_log(slog.isEnabledFor(logging.INFO) and not args.until), smooth.Smoother(mesh, inp.solver, inp.wall_control_function) as sm:
        slog.info("hip solver, inner strategy: %s%s", sm.inner.name, " (chosen from the block sizes and the spread of the cells' aspect ratios)" if inp.solver.inner == solver.Inner.auto else "")
        if args.until:
            reached, stats = sm.iterate_until(args.until, iterations or 100)
            slog.info("scaled residual %.3e after %d iterations (%s)", stats["scaled_residual_rms"], stats["outer_iterations"], "reached" if reached else "NOT reached")
        else:
            stats = sm.iterate(iterations)
            if stats["not_converged"]:
                slog.warning("hip solve did not converge in %d of %d outer iterations", stats["not_converged"], stats["outer_iterations"])
            slog.info("elapsed time for smoothing: %.2f s", stats["seconds"])
        sm.download()
        stats["inner"] = sm.inner.name
    out = args.output or inp.output
    if out:
        mesh.write(out)
        logging.getLogger("output").info("wrote %s (%d blocks, %d nodes)", out, len(mesh.blocks), sum(b.points.data.shape[0] * b.points.data.shape[1] for b in mesh.blocks))
    return stats


if __name__ == "__main__":
    main()
