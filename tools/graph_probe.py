#!/usr/bin/env python3
"""Why is hipGraph replay of the training step slower than eager launching?

  python tools/graph_probe.py [steps=20]

Times the configs[1] step four ways in ONE process — eager / graph replay, each with the three-stream schedule and with
everything on one stream (GLOWTTS_SIDE_STREAM=0) — and dumps the captured graphs (hipGraphDebugDotPrint): nodes, edges, how
many nodes have more than one predecessor / successor (the cross-stream joins), the length of the longest dependency chain.
If replay(3 streams) ~ replay(1 stream) ~ eager(1 stream), the graph executor is running the captured branches one after
another: the step's 18 % gain from overlapping streams is what replay loses."""
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "glow-tts-train_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402

import bench  # noqa: E402


def dot_stats(path):
    txt = open(path).read()
    edges = re.findall(r'"?([\w.]+)"?\s*->\s*"?([\w.]+)"?', txt)
    nodes = set(re.findall(r'^\s*"?([\w.]+)"?\s*\[', txt, flags=re.M))
    for a, b in edges:
        nodes.add(a)
        nodes.add(b)
    preds, succs = {}, {}
    for a, b in edges:
        preds.setdefault(b, []).append(a)
        succs.setdefault(a, []).append(b)
    depth = {}
    order = list(nodes)
    # longest chain by repeated relaxation over a topological order (Kahn)
    indeg = {n: len(preds.get(n, [])) for n in nodes}
    ready = [n for n in nodes if indeg[n] == 0]
    while ready:
        n = ready.pop()
        depth[n] = 1 + max((depth[q] for q in preds.get(n, [])), default=0)
        for m in succs.get(n, []):
            indeg[m] -= 1
            if indeg[m] == 0:
                ready.append(m)
    kinds = {}
    for label in re.findall(r'label="([^"]*)"', txt):
        k = label.split("\\n")[0].split("(")[0].strip()[:40]
        kinds[k] = kinds.get(k, 0) + 1
    return {"nodes": len(nodes), "edges": len(edges), "joins(>1 pred)": sum(len(v) > 1 for v in preds.values()),
            "forks(>1 succ)": sum(len(v) > 1 for v in succs.values()), "longest_chain": max(depth.values(), default=0),
            "roots": sum(1 for n in nodes if not preds.get(n)), "top_labels": sorted(kinds.items(), key=lambda kv: -kv[1])[:6]}


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    sys.argv = [sys.argv[0]]                        # bench.parse() reads the defaults: BASELINE configs[1]
    args = bench.parse()
    dev = torch.device("cuda", 0)
    from glow_tts_train import _hip
    from glow_tts_train.train import train_batch

    _hip.load()
    model, opt, batch, cfg = bench.build_workload(args, dev, 0)
    step = lambda: train_batch(model, opt, batch, cfg.grad_clip, None)      # noqa: E731

    def timed(fn, n=steps):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n

    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    for streams in ("3", "1"):
        os.environ["GLOWTTS_SIDE_STREAM"] = "1" if streams == "3" else "0"
        eager = timed(step)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                step()
        torch.cuda.current_stream().wait_stream(side)
        host_state = (opt.step_num, opt.cur_lr)
        g = torch.cuda.CUDAGraph()
        g.enable_debug_mode()
        t0 = time.perf_counter()
        with torch.cuda.graph(g):
            step()
        cap_ms = 1e3 * (time.perf_counter() - t0)
        opt.step_num, opt.cur_lr = host_state
        dot = os.path.join(out_dir, f"graph_{streams}stream.dot")
        g.debug_dump(dot)
        replay = timed(g.replay)
        print(f"{streams} stream(s): eager {eager:6.2f} ms/step   graph replay {replay:6.2f} ms/step   (capture {cap_ms:.0f} ms)", flush=True)
        try:
            print("   graph:", dot_stats(dot), flush=True)
        except Exception as exc:
            print("   dot parse failed:", exc)
        del g


if __name__ == "__main__":
    main()
