#!/bin/bash
set +e
bash scripts/gpu_tests.sh r2s9
bash scripts/gpu_profile_r2.sh r2s9prof
