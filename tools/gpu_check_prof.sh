#!/bin/bash
bash tools/gpu_check.sh || true
bash tools/gpu_prof.sh | grep -E "finish|point_kernel|integrate|dehoog"
