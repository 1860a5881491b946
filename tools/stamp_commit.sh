#!/bin/bash
# Leaves the current commit in .commit_stamp (git-ignored, travels with gpurun): the GPU box has no .git, and
# bench.py / tools/pmc_traffic.py stamp what they measure with it.  Installed as .git/hooks/post-commit too.
cd "$(git rev-parse --show-toplevel)" && git rev-parse --short HEAD > .commit_stamp
