"""Batch driver with the reference's command line and CSV schema (``/root/reference/PDLP/main.py:11-174``):

    python -m torchpdlp_amd --instance_path DIR [--tolerance 1e-4] [--output_path output] [--precondition]
        [--primal_weight_update] [--adaptive_stepsize] [--verbose] [--max_kkt N] [--time_limit S]

Solves every ``*.mps`` in the folder (sorted), continues past failures and writes ``solver_results.csv`` with the
columns ``File, Objective, Iterations (k), Restarts (n), KKT Passes (j), Time (s), Status``.
"""
import argparse
import csv
import os
import sys

import torch

from .api import solve_lp
from .mps import mps_to_standard_form

COLUMNS = ["File", "Objective", "Iterations (k)", "Restarts (n)", "KKT Passes (j)", "Time (s)", "Status"]


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Run the MI355X PDLP solver over a folder of MPS instances.")
    p.add_argument("--device", type=str, choices=["cpu", "gpu", "auto"], default="auto",
                   help="kept for compatibility; the solver runs on the HIP device (there is no CPU path)")
    p.add_argument("--instance_path", type=str, default="feasible")
    p.add_argument("--tolerance", type=float, default=1e-4)
    p.add_argument("--output_path", type=str, default="output")
    p.add_argument("--precondition", action="store_true")
    p.add_argument("--primal_weight_update", action="store_true")
    p.add_argument("--adaptive_stepsize", action="store_true")
    p.add_argument("--adaptive_retry", action="store_true",
                   help="with --adaptive_stepsize: repeat a rejected step with the shrunk step size until one is accepted (the loop "
                        "the reference's adaptive step was meant to be, enhancements/test_ass.py:322-363); default: one trial, as the "
                        "live package does")
    p.add_argument("--direct_exchange", action="store_true",
                   help="several ranks of one node (not in the reference's CLI): iterate without collectives -- every half-step stores "
                        "its block straight into the other ranks' memory over HIP IPC / xGMI; cross-checked against the "
                        "collective-driven loop first, which stays in charge if anything differs")
    p.add_argument("--infeasibility_detect", action="store_true")
    p.add_argument("--verbose", action="store_true")
    p.add_argument("--support_sparse", action="store_true", help="accepted; matrices are always sparse here")
    p.add_argument("--max_kkt", type=int, default=100_000)
    p.add_argument("--time_limit", type=int, default=3600)
    p.add_argument("--fishnet", action="store_true")
    p.add_argument("--dtype", choices=["fp32", "fp64", "mixed"], default="fp32",
                   help="for tolerances below float32 resolution: mixed (float32 matrix entries under float64 vectors, the fast way) or fp64")
    p.add_argument("--seed", type=int, default=None, help="pins the power-iteration start vector (unseeded in the reference)")
    p.add_argument("--standard_mps", action="store_true", help="standard meaning of FR/MI/PL/BV bounds instead of the reference's")
    return p.parse_args(argv)


def _fail_row(name, what, e):
    msg = str(e)
    status = f"{what}: {msg[:50]}..." if len(msg) > 50 else f"{what}: {msg}"     # main.py:101,161
    return {"File": name, "Objective": "N/A", "Iterations (k)": "N/A", "Restarts (n)": "N/A", "KKT Passes (j)": "N/A",
            "Time (s)": "N/A", "Status": status}


def main(argv=None) -> int:
    args = parse_args(argv)
    if args.device == "cpu" or not torch.cuda.is_available():
        print("torchpdlp_amd needs a HIP device: there is no CPU solver path in this package.", file=sys.stderr)
        return 2
    # under torchrun (one process per GPU) every instance is sharded over the ranks; rank 0 reports
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    comm = None
    if world > 1:
        import torch.distributed as dist
        # (PDLP_SHARE_GPU=1 + PDLP_DIST_BACKEND=gloo: rehearsal of the sharded path on a one-GPU box)
        torch.cuda.set_device(0 if os.environ.get("PDLP_SHARE_GPU") == "1" else int(os.environ.get("LOCAL_RANK", "0")))
        if not dist.is_initialized():
            dist.init_process_group(os.environ.get("PDLP_DIST_BACKEND", "nccl"))
        comm = True
    else:
        dist = None
    if rank == 0:
        print(f"PyTorch is using ROCm/CUDA device: {torch.cuda.get_device_name(torch.cuda.current_device())}"
              + (f" x {world} ranks" if world > 1 else ""))
    files = sorted(f for f in os.listdir(args.instance_path) if f.endswith(".mps"))       # main.py:83
    results = []

    agree_timeout = float(os.environ.get("PDLP_FAIL_AGREE_TIMEOUT", "60"))

    def failed_ranks(failed: bool, tag: str) -> int:
        """how many ranks failed on this instance (-1: no agreement in time); out of band, never a collective: distributed.agree_failed"""
        if world == 1:
            return int(failed)
        from .distributed import agree_failed
        return agree_failed(dist, rank, world, failed, tag, agree_timeout)

    def write_results():
        os.makedirs(args.output_path, exist_ok=True)
        if results:
            out = os.path.join(args.output_path, "solver_results.csv")
            with open(out, "w", newline="") as f:
                w = csv.DictWriter(f, fieldnames=COLUMNS)
                w.writeheader()
                w.writerows(results)
            print(f"Results saved to CSV instead: {out}")
        else:
            print("No results to save.")
    for name in files:
        path = os.path.join(args.instance_path, name)
        print(f"\nProcessing {path}...")
        dtype = torch.float32 if args.dtype == "fp32" else torch.float64
        try:                            # main.py:88-103: a file that does not load gets its own kind of row
            # (sharded runs parse on the host and put only this rank's blocks on its GPU: api._solve_lp_sharded)
            problem = mps_to_standard_form(path, device="cpu" if world > 1 else torch.device("cuda", torch.cuda.current_device()),
                                           verbose=args.verbose, compat=not args.standard_mps, dtype=dtype)
        except Exception as e:
            print(f"Failed to load MPS file: {path}. Error: {e}")
            msg = str(e)
            row = _fail_row(name, "Failed to load", e)
            row["Status"] = f"Failed to load: {msg[:50]}..." if len(msg) > 50 else msg        # (the reference's quirk: main.py:101)
            results.append(row)
            continue
        try:
            r = solve_lp(problem, tol=args.tolerance, precondition=args.precondition, primal_weight_update=args.primal_weight_update,
                         adaptive_stepsize=args.adaptive_stepsize, adaptive_retry=args.adaptive_retry, max_kkt=args.max_kkt, time_limit=args.time_limit,
                         verbose=args.verbose, dtype=dtype, seed=args.seed, fishnet=args.fishnet, comm=comm,
                         infeasibility_detect=args.infeasibility_detect, precision="mixed" if args.dtype == "mixed" else None,
                         direct_exchange=args.direct_exchange)
            print(f"Solver uses {r.time:.4f} seconds.\nStatus: {r.status}")
            results.append({"File": name, "Objective": f"{r.objective:.6f}", "Iterations (k)": r.iterations, "Restarts (n)": r.restarts,
                            "KKT Passes (j)": r.kkt_passes, "Time (s)": f"{r.time:.4f}", "Status": r.status})      # main.py:142-150
            failed = False
        except Exception as e:          # the reference records the failure and goes on (main.py:152-162)
            print(f"Solver failed for {name}. Error: {e}")
            results.append(_fail_row(name, "Solver failed", e))
            failed = True
        # (outside the try: a store error in the agreement itself must not add a second row for the instance)
        bad = failed_ranks(failed, name)
        if bad != 0 and bad != world:
            # only some ranks failed (out of memory on one shard, ...), or the ranks could not agree in time: they are out of step and
            # the next instance would hang in its first collective.  Rank 0 saves what there is -- the rows of the instances solved so
            # far -- and the job ends non-zero.  (A failure every rank hits alike -- an unsupported option combination, a bad
            # instance -- is a row like any other: the loop goes on, as the reference's does.)
            print(f"rank {rank}: " + (f"{bad} of {world} ranks failed" if bad > 0 else f"no agreement within {agree_timeout:.0f} s") +
                  f" on {name}; leaving the sharded run", file=sys.stderr, flush=True)
            if rank == 0:
                if results[-1]["Status"] == "Solved" or not results[-1]["Status"].startswith("Solver failed"):
                    results[-1] = _fail_row(name, "Solver failed", RuntimeError(f"{bad} of {world} ranks failed" if bad > 0 else "ranks out of step"))
                write_results()
            sys.stdout.flush()
            os._exit(3)
    if rank != 0:
        return 0
    write_results()
    return 0


if __name__ == "__main__":
    sys.exit(main())
