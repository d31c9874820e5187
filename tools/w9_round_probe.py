"""Latency of ONE wave-wide MiMC7 round on a lone wave (og_ubench_cycles kinds 200 + FORM, hooks build): FORM 0 one row, four products
deep; 1 two rows, three deep; 2 two rows + the 32-bit Montgomery digit, per row in every product.  Kinds 210 + 4 D + 2 S + G take a
product apart (three products per iteration): D = the digit on the scalar unit | per row through DPP | none, S = the lane shift with |
without its DPP move, G = without | with the gather.  Prints cycles per round (per iteration).   usage: w9_round_probe.py [kinds ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OWSHEN_GPU_LIB", os.path.join(ROOT, "owshen_amd", "libowshen_gpu_hooks.so"))
from owshen_amd import api  # noqa: E402


def main():
    ctx = api.Context(0)
    iters = 2000
    kinds = [int(a) for a in sys.argv[1:]] or [200, 201, 202]
    for rep in range(2):
        for kind in kinds:
            try:
                ms, cyc = ctx.ubench_cycles(kind, iters, 1)
            except Exception as e:  # noqa: BLE001
                print(kind, "failed", e)
                continue
            print(f"kind {kind}: {cyc / iters:8.1f} cycles per round, {ms * 1e3 / iters:7.3f} us per round")


if __name__ == "__main__":
    main()
