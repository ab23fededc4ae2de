"""Per-forecast cost of an autoregressive rollout (LatentSpaceAutoregressive.autoregressive_sample calls `sample` once per forecast
step, autoregressivesample.py:83-183): ONE captured replay per forecast on a small latent, so what the host does around the
replay -- plan key, load, result copy, range guard -- is a visible share.  Prints seconds per forecast (wall, synchronised per
forecast as the rollout is: each forecast's result conditions the next) and the host time of one call with the device idle.
Run from the tree to measure (`python tools/forecast_time.py`); tools/ab_bench.sh-style A/B: run it in ab_r2/ and here."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels", type=int, default=32)
    ap.add_argument("--size", type=int, default=32)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--nsteps", type=int, default=10)
    ap.add_argument("--forecasts", type=int, default=200)
    a = ap.parse_args()
    import diffsci_amd.models as M
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = M.PUNetG(M.PUNetGConfig(model_channels=a.channels, input_channels=4, output_channels=4))
    module = M.KarrasModule(net, M.KarrasModuleConfig.from_edm()).to(dev).eval()
    x = torch.randn(a.batch, 4, a.size, a.size, device=dev)
    with torch.inference_mode():
        for _ in range(3):
            y = module.propagate_white_noise(x, nsteps=a.nsteps)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.forecasts):
            y = module.propagate_white_noise(x, nsteps=a.nsteps)
            x = torch.roll(y, 1, 0)                 # the next forecast is conditioned on this one: a dependency, not idle time
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / a.forecasts
        # the replay alone, back to back: what the device needs for one forecast
        loop, g = next(iter(module._plans.plans.values()))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(module._plans.stream):
            for _ in range(a.forecasts):
                g.launch()
        torch.cuda.synchronize()
        replay = (time.perf_counter() - t0) / a.forecasts
    print(json.dumps({"workload": f"PUNetG-{a.channels} [{a.batch},4,{a.size},{a.size}] {a.nsteps}-step Heun per forecast",
                      "ms_per_forecast": round(wall * 1e3, 4), "ms_replay_only": round(replay * 1e3, 4),
                      "ms_around_the_replay": round((wall - replay) * 1e3, 4), "tree": os.path.basename(os.getcwd())}))


if __name__ == "__main__":
    main()
