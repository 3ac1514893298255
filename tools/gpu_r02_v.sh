#!/bin/bash
timeout 900 python -m pytest tests/test_tube.py tests/test_hip_ops.py tests/test_step.py -q -p no:cacheprovider 2>&1 | tail -4 | cut -c1-300
