"""Trainer.train(epochs) through the drop-in module with the epochs enqueued ahead of the loss read-back (default) and
without (GM_PIPELINE_EPOCHS=0), alternating in one process: us per step, as bench.py's `config.trainer` measures it."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("gm_bench_ab", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)

out = []
for rep in range(3):
    for pipe in ("1", "0"):
        os.environ["GM_PIPELINE_EPOCHS"] = pipe
        for epochs in (3, 10):
            r = bench.bench_trainer(epochs)
            out.append(dict(pipelined=pipe == "1", epochs=epochs, us_per_step=round(r["ms_per_step"] * 1e3, 2)))
            print(out[-1], flush=True)
json.dump(out, open(sys.argv[1], "w"), indent=1) if len(sys.argv) > 1 else None
