"""The multi-GPU bench entry point with the REAL backend (nccl = RCCL) on one rank: exercises process-group
initialisation with device_id, barrier and the all-reduces on this box (no neighbour exchange with one rank)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29551")
os.environ["SPH_BENCH_BACKEND"] = "nccl"
from sph_amd import slab
args = argparse.Namespace(gpus=1, steps=5, warmup=5, particles=int(sys.argv[1]) if len(sys.argv) > 1 else 1048576,
                          workload="", no_cpu_baseline=True)
slab.bench_main(args, 0, 0, 1)
