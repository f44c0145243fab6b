#!/bin/bash
# optimize_contrast(optimizer='evk_bfgs'): the library's loop against the Python loop, three sizes
f() { grep "^n="; }
for spec in "10000000 480 640" "1000000 480 640" "50000000 720 1280"; do
  for py in 0 1; do
    echo "python_loop=$py"; EVK_BFGS_PYTHON_LOOP=$py python tools/bfgs_profile.py $spec 2>&1 | f
  done
done
