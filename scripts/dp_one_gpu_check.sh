#!/bin/bash
# Exercises the multi-process launch paths (bench.py --gpus N, run.py --gpus N --jobs) on a ONE-GPU box: gloo backend, every
# local rank on device 0 (ALG_DIST_BACKEND / ALG_DIST_ONE_GPU test hooks).  Throughput is meaningless here (the ranks share the
# GPU); what is checked is the self-launch, the CUDA-tensor weight broadcast, the barriers / max-over-ranks and bit-identity of
# the data-parallel outputs with single-process runs.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O /tmp/dp
export ALG_DIST_BACKEND=gloo ALG_DIST_ONE_GPU=1 PYTHONPATH=$R
cd $R
timeout 900 python bench.py --gpus 2 --layers 4 --steps 3 --warmup 1 --no-cpu-baseline > $O/dp2_bench.json 2> $O/dp2_bench.err
echo "bench --gpus 2 exit $?"; python - <<PY
import json
d = json.loads(open("$O/dp2_bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("n_gpus", "value", "ms_per_step", "scaling")}, d["config"]["parallelism"])
assert d["n_gpus"] == 2
PY
# the other multi-GPU workloads of BASELINE.json: config 4 = HunyuanVideo, 8-way data parallel (here 2 ranks,
# weights through the RCCL-shaped broadcast), config 5 = Wan 14B fp8 with the CFG pair on a GPU pair (one all-gather per step)
for leg in "c4 --gpus 2" "c5 --gpus 2 --cfg-split" "c3 --gpus 2"; do
  tag=$(echo $leg | tr -d ' -')
  timeout 1200 python bench.py --workload $leg --layers 2 --steps 2 --warmup 1 --no-cpu-baseline > $O/dp2_$tag.json 2> $O/dp2_$tag.err
  echo "bench --workload $leg exit $?"; python - <<PY
import json
d = json.loads(open("$O/dp2_$tag.json").read().strip().splitlines()[-1])
print("$leg", {k: d[k] for k in ("n_gpus", "value", "ms_per_step", "scaling")}, d["config"]["parallelism"])
assert d["n_gpus"] == 2 and d["finite"]
PY
done
cat > /tmp/dp/c.yaml <<Y
model: {path: CogVideoX-toy, dtype: bfloat16, synthetic_config: {num_layers: 2, num_attention_heads: 8, text_embed_dim: 128, max_text_seq_length: 16, sample_height: 8, sample_width: 12, sample_frames: 9, time_embed_dim: 64}}
generation: {num_inference_steps: 3, height: 64, width: 96, num_frames: 9, guidance_scale: 6.0}
alg: {use_low_pass_guidance: true, lp_filter_type: down_up, lp_filter_in_latent: true, lp_resize_factor: 0.25, lp_strength_schedule_type: interval, schedule_interval_end_time: 0.4}
video: {fps: 8}
Y
python - <<PY
import yaml
yaml.safe_dump([{"output_path": "/tmp/dp/dp_%d.pt" % v} for v in range(3)], open("/tmp/dp/jobs.yaml", "w"))
yaml.safe_dump([{"output_path": "/tmp/dp/solo_%d.pt" % v} for v in range(3)], open("/tmp/dp/solo.yaml", "w"))
PY
timeout 600 python run.py --config /tmp/dp/c.yaml --synthetic --gpus 2 --jobs /tmp/dp/jobs.yaml > $O/dp2_run.log 2>&1; echo "run.py --gpus 2 exit $?"
env -u ALG_DIST_BACKEND -u ALG_DIST_ONE_GPU timeout 600 python run.py --config /tmp/dp/c.yaml --synthetic --jobs /tmp/dp/solo.yaml > $O/dp1_run.log 2>&1; echo "run.py single exit $?"
python - <<PY
import torch
for v in range(3):
    a, b = torch.load("/tmp/dp/dp_%d.pt" % v), torch.load("/tmp/dp/solo_%d.pt" % v)
    assert torch.equal(a, b), v
    print("job", v, tuple(a.shape), "data-parallel == single process")
PY
tail -3 $O/dp2_run.log
