#!/bin/bash
# round 5: weight preparation rows kept in registers between the maximum pass and the split pass -- tests of the prepared forms, then the batched kernel inside the step
python -m pytest tests/test_conv_split.py tests/test_mixed.py tests/test_upconv_bf16.py -q -m gpu -x 2>&1 | tail -2
bash tools/gpu_r05_instep_ab.sh "prepare_weights\|conv_split_prepare\|upconv_split_prepare"
