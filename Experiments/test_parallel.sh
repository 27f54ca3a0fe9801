#!/bin/bash
# Kept under the reference's name for its README command (`./test_parallel.sh --dataset B ...`); everything happens in
# `python -m test launch`: one rank per GPU (LIDARREG_GPUS="0 1 ..." picks devices), every rank watched, analysis only if all succeed.
PYTHONPATH="$(cd "$(dirname "$0")" && pwd)${PYTHONPATH:+:$PYTHONPATH}" exec python -m test launch "$@"
