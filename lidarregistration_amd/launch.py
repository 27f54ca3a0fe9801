"""One OS process per GPU: start the ranks, watch ALL of them, stop everything when one fails.

The reference starts its ranks from a shell loop and waits for all of them (Experiments/test_parallel.sh:18-22): a rank that dies goes
unnoticed, the others run to the end and the analysis pass reads partial files.  Here the parent polls every child; the first non-zero
exit ends the run at once and the remaining ranks are killed.  The parent never touches HIP (no torch import in this module), so
starting children after it is safe on this platform -- gpu_list() counts devices from sysfs or the inherited *_VISIBLE_DEVICES, never through HIP.  Used by ``bench.py --gpus N`` (self-launch) and by the CLI's
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


KFD_NODES = "/sys/class/kfd/kfd/topology/nodes"


def kfd_gpu_count(root=KFD_NODES):
    """GPUs the kernel driver exposes, counted WITHOUT a HIP call: a topology node is a GPU when its `properties` file has
    simd_count > 0 (CPU nodes have 0).  None when the directory cannot be read."""
    try:
        nodes = sorted(os.listdir(root), key=lambda v: (len(v), v))
    except OSError:
        return None
    n = 0
    for node in nodes:
        try:
            with open(os.path.join(root, node, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
        except OSError:
            continue          # (a node this user may not read: the driver hides GPUs outside the container's cgroup that way)
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    return n


def gpu_list(environ=None, kfd_root=KFD_NODES):
    """One entry per rank: the environment variable (and its value) that pins the rank to its GPU -- decided without initialising HIP in
    this process (the launcher starts children and later runs the analysis pass itself):
      1. LIDARREG_GPUS="0 1 2": HIP_VISIBLE_DEVICES = each entry as given (repeats allowed: several ranks on one GPU);
      2. else the entries of an inherited HIP_VISIBLE_DEVICES, ROCR_VISIBLE_DEVICES or CUDA_VISIBLE_DEVICES (first one set): they already
         are the ids (or UUIDs) of the devices this job may use, so each rank gets ONE of them under the SAME variable -- not a fresh
         0..n-1 numbering, which would name other physical devices;
      3. else one rank per GPU node of the KFD topology in sysfs (HIP_VISIBLE_DEVICES = 0 .. n-1);
      4. else (no sysfs) as many as a short-lived child process counts with torch.cuda.device_count()."""
    env = os.environ if environ is None else environ
    given = env.get("LIDARREG_GPUS")
    if given:
        return [("HIP_VISIBLE_DEVICES", v) for v in given.replace(",", " ").split()]
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        if env.get(var, "").strip():
            return [(var, v.strip()) for v in env[var].split(",") if v.strip()]
    n = kfd_gpu_count(kfd_root)
    if n is None:
        import sys
        try:
            out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True, timeout=300)
            n = int(out.stdout.strip().splitlines()[-1])
        except Exception:
            n = 0
    return [("HIP_VISIBLE_DEVICES", str(i)) for i in range(n)]
