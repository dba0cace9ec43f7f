#!/usr/bin/env python3
"""Generate tests/golden/oracle_fp32_noise.json: the REFERENCE ARITHMETIC's own gradient error, per tensor.

The teacher-forced GPU tests (tests/test_model_gpu.py::_teacher_forced) bound every gradient tensor of the HIP path by
max(1e-4, 4 x the error the oracle's fp32 CPU run makes against an fp64 run on ITS OWN activation sign pattern).  That error
depends on the oracle alone (oracle/discogan_ref.py + tests/kink_probe.py -- no reference code, no GPU), and computing it
live costs one more fp64 CPU pass of all four networks per iteration: 150-250 s of the GPU suite on the GPU box's host cores.
This script computes it once, here; the tests read the numbers (DG_LIVE_NOISE=1 recomputes them live instead).

    python tests/golden/make_oracle_noise.py [--skip512]

Runs: "SxN" = the standard sequential run of the tests (seed-1234 state, synthetic batch seed 0, iterations 0..3, Adam on the
oracle's own gradients between them) for 16x4, 64x4, 128x2 and 512x2 (test_teacher_forced_iterations_512_f32x3);
"512x2_init" = iterations 0 (D-step) and 1 (G-step) from the seeded init without an optimiser step
(test_masked_fp64_gradient_parity_512).  Values: relative L2 error of the fp32
gradient against the fp64 gradient on the same piecewise-linear function.  Data only.
"""
import argparse
import json
import os
import sys
import time

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

from oracle import discogan_ref as O  # noqa: E402
import kink_probe as KP  # noqa: E402


def run(S, N, iters, step):
    st = O.build_state(image_size=S, seed=1234)
    A, B = O.synthetic_batch(N, S, seed=0)
    out = {}
    for it in iters:
        t0 = time.time()
        with KP.record_masks_oracle(st.nets) as m32:
            O.train_iteration(st, A, B, it, do_step=False)
        s_o = KP.run_masked64(O, st, m32, A, B, it)
        dstep = O.is_dis_step(it, st.args)
        live = ("dis_A", "dis_B") if dstep else ("gen_A", "gen_B")
        rows = {}
        for name in live:
            po, to = dict(st.nets[name].named_parameters()), dict(s_o.nets[name].named_parameters())
            for pn in po:
                rows[f"{name}.{pn}"] = KP.rel_err(po[pn].grad, to[pn].grad)
        out[str(it)] = rows
        print(f"S={S} N={N} iter {it} ({'D' if dstep else 'G'}): {len(rows)} tensors, worst {max(rows.values()):.2e}, "
              f"{time.time() - t0:.0f} s", flush=True)
        if step:
            (st.optim_dis if dstep else st.optim_gen).step()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip512", action="store_true")
    ap.add_argument("--only512", action="store_true")
    a = ap.parse_args()
    path = os.path.join(HERE, "oracle_fp32_noise.json")
    runs = {}
    if os.path.exists(path):
        runs = json.load(open(path))["runs"]
    if not a.only512:
        for S, N in ((16, 4), (64, 4), (128, 2)):
            runs[f"{S}x{N}"] = run(S, N, range(4), True)
    if not a.skip512:
        runs["512x2_init"] = run(512, 2, [0, 1], False)
        runs["512x2"] = run(512, 2, range(4), True)
    meta = {"source": "oracle/discogan_ref.py fp32 vs fp64 on its own activation pattern (tests/kink_probe.py)",
            "torch": torch.__version__, "threads": torch.get_num_threads(), "seed_model": 1234, "seed_data": 0}
    with open(path, "w") as f:
        json.dump({"meta": meta, "runs": runs}, f, indent=0, sort_keys=True)
    print("wrote", path)


if __name__ == "__main__":
    main()
