#!/bin/bash
bash tools/gpu_check.sh || true
for p in 16 32 64; do echo "== UCF_FINISH_PART=$p"; UCF_FINISH_PART=$p bash tools/gpu_prof.sh | grep -E "finish|point_kernel|integrate|dehoog"; done
