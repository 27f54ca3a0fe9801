"""One OS process per GPU: start the ranks, watch ALL of them, stop everything when one fails.

The reference starts its ranks from a shell loop and waits for all of them (Experiments/test_parallel.sh:18-22): a rank that dies goes
unnoticed, the others run to the end and the analysis pass reads partial files.  Here the parent polls every child; the first non-zero
exit ends the run at once and the remaining ranks are killed.  The parent never touches HIP (no torch import in this module), so
starting children after it is safe on this platform.  Used by ``bench.py --gpus N`` (self-launch) and by the CLI's
``python -m test launch`` (= ``test_parallel.sh``).
"""
import os
import socket
import subprocess
import time


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def run_ranks(commands, envs=None, poll_s=0.05, cwd=None):
    """Start one child per entry of `commands` (argv lists; `envs[i]` is merged over os.environ) and wait.
    Returns 0 when every child exited with 0, else the first non-zero exit code seen (as a positive number; the rest are killed)."""
    procs = []
    for i, cmd in enumerate(commands):
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if envs is not None:
            env.update(envs[i])
        procs.append(subprocess.Popen(cmd, env=env, cwd=cwd))
    rc = 0
    try:
        live = list(procs)
        while live and rc == 0:
            for p in list(live):
                r = p.poll()
                if r is not None:
                    live.remove(p)
                    rc = max(rc, abs(r))
            if live and rc == 0:
                time.sleep(poll_s)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            try:
                p.wait(timeout=10)
            except Exception:
                pass
    return rc


def gpu_list():
    """Device ids to start one rank on: LIDARREG_GPUS="0 1 2" (repeats allowed: several ranks on one GPU), else every visible device.
    (torch.cuda.device_count() does not initialise HIP on this platform.)"""
    given = os.environ.get("LIDARREG_GPUS")
    if given:
        return [int(v) for v in given.replace(",", " ").split()]
    import torch
    return list(range(torch.cuda.device_count()))
