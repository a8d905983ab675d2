"""Probe: does running two independent half-batch training steps on two HIP streams beat one full-batch step?
(kernel tails / ramp-ups of one stream filled by the other).  Developer experiment, not part of the product."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench                                                  # noqa: E402
from constant_memory_waveglow_amd.parallel import FlowTrainer  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    B = 24
    full = FlowTrainer(bench.build_model(dev), bench.SIGMA)
    halves = [FlowTrainer(bench.build_model(dev), bench.SIGMA) for _ in range(2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    x = torch.rand(B, bench.SEG, device=dev) * 2 - 1
    h = torch.randn(B, 80, bench.FRAMES, device=dev)
    xs, hs = x.chunk(2), h.chunk(2)

    def step_full():
        full.step(x, h)

    def step_two():
        for i in range(2):
            with torch.cuda.stream(streams[i]):
                halves[i].step(xs[i], hs[i])

    for fn, name in ((step_full, "one stream, B=24"), (step_two, "two streams, 2 x B=12")):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print("%-24s %.2f ms/step  %.3f M samples/s" % (name, dt * 1e3, B * bench.SEG / dt / 1e6), flush=True)


if __name__ == "__main__":
    main()
