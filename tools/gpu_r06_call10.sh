#!/bin/bash
for i in 1 2 3 4 5 6; do timeout 300 python tools/r06_diag_headconv3.py 2>&1 | tail -6 | cut -c1-900; done
