#!/bin/bash
# round 5: seg_mean3_maxlabel with four points per lane in flight (same add order) -- tests, then inside the step against the previous library
python -m pytest tests/test_hip_ops.py tests/test_prepare_batch.py -q -m gpu -x -k "seg or mean or pillar or prepare or collate" 2>&1 | tail -2
bash tools/gpu_r05_instep_ab.sh "seg_mean3\|offset_centres\|pfn_features"
