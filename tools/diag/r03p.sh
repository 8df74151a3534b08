#!/bin/bash
tools/ab_kernel.sh 2 paired paired_noatom -- --no-stage-rooflines --no-renderer-only
