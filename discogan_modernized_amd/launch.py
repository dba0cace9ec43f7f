"""Single-node launcher (reference: distributed_training.sh:1-117).

Same flag surface as the reference's shell launcher (``--task_name= --model_arch= --batch_size= --epochs=
--learning_rate= --image_size= --style_A= --style_B= --gpus=``; ``--gpus`` is a comma-separated device list whose
length is the world size, :5-8,69-73) but the world size comes from the list / the environment instead of the
hard-coded defaults, the rendezvous address is 127.0.0.1 and one process per GPU is started through
``python -m torch.distributed.run`` (the reference uses the deprecated torch.distributed.launch, :100-114).
Anything the launcher does not know is passed through to distributed_image_translation.py.

    python -m discogan_modernized_amd.launch --gpus=0,1,2,3,4,5,6,7 --task_name=celebA --style_A=Male \\
        --style_B=Smiling --batch_size=64 --image_size=64

The launcher itself never touches the GPU (it only counts devices), so spawning children from it is safe.
"""
from __future__ import annotations

import os
import subprocess
import sys
from datetime import datetime

KNOWN = ("task_name", "model_arch", "batch_size", "epochs", "learning_rate", "image_size", "style_A", "style_B")
DEFAULTS = dict(task_name="edges2shoes", model_arch="discogan", batch_size="64", epochs="50", learning_rate="0.0002",
                image_size="64")            # distributed_training.sh:10-17


def build_command(argv, environ=None):
    """Returns (cmd list, env dict, log path or None).  Pure function (tested on CPU)."""
    env = dict(environ if environ is not None else os.environ)
    opts, extra, gpus, port, log_dir = dict(DEFAULTS), [], None, env.get("MASTER_PORT", "29500"), None
    for a in argv:
        if a.startswith("--gpus="):
            gpus = a.split("=", 1)[1]
        elif a.startswith("--master_port="):
            port = a.split("=", 1)[1]
        elif a.startswith("--log_dir="):
            log_dir = a.split("=", 1)[1]
        elif a.startswith("--") and "=" in a and a[2:].split("=", 1)[0] in KNOWN:
            k, v = a[2:].split("=", 1)
            opts[k] = v
        else:
            extra.append(a)
    if gpus is None:
        gpus = env.get("HIP_VISIBLE_DEVICES") or env.get("CUDA_VISIBLE_DEVICES")
    if gpus is None:
        import torch
        # Counting devices may initialise the HIP runtime in THIS process (it can on ROCm builds).  That is harmless here only
        # because the ranks are started as CHILD processes below (subprocess), never by replacing this process with exec.
        gpus = ",".join(str(i) for i in range(max(torch.cuda.device_count(), 1)))
    world = len([g for g in gpus.split(",") if g != ""])
    env["HIP_VISIBLE_DEVICES"] = gpus
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ts = datetime.now().strftime("%Y%m%d_%H%M%S")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           "-m", "discogan_modernized_amd.distributed_image_translation", "--distributed", f"--world_size={world}",
           f"--results_dir=./results/distributed_{opts['task_name']}_{opts['model_arch']}_{ts}",
           f"--models_dir=./models/distributed_{opts['task_name']}_{opts['model_arch']}_{ts}"]
    for k in KNOWN:
        if k in opts and opts[k] != "":
            cmd.append(f"--{k}={opts[k]}")
    cmd += extra
    log = os.path.join(log_dir, "train.log") if log_dir else None
    return cmd, env, log


def main(argv=None):
    cmd, env, log = build_command(sys.argv[1:] if argv is None else argv)
    print("launching:", " ".join(cmd), flush=True)
    if log:
        os.makedirs(os.path.dirname(log), exist_ok=True)
        with open(log, "w") as f:
            return subprocess.call(cmd, env=env, stdout=f, stderr=subprocess.STDOUT)
    return subprocess.call(cmd, env=env)


if __name__ == "__main__":
    sys.exit(main())
