#! /bin/bash
# One process per GPU, pairs sharded round-robin, results merged by the analysis pass
# (same protocol as the reference's Experiments/test_parallel.sh; GPUs are enumerated through torch instead of nvidia-smi).
if [ -n "$LIDARREG_GPUS" ]; then
	gpu_inds=( $LIDARREG_GPUS )
else
	n_gpus_in_system=$(python -c "import torch; print(torch.cuda.device_count())")
	let max_gpu_ind=$n_gpus_in_system-1
	gpu_inds=($(seq 0 $max_gpu_ind))
fi
n_gpus=${#gpu_inds[@]}

file_base=$(mktemp)
start_time=$(date '+%Y%m%d_%H_%M_%S')
for i in ${!gpu_inds[@]}; do
  HIP_VISIBLE_DEVICES=${gpu_inds[$i]} python -m test test_parallel $start_time $file_base $n_gpus ${i} "$@" &
done

wait < <(jobs -p)

python -m test test_parallel $start_time $file_base $n_gpus analysis "$@"
