#!/bin/bash
tools/ubench/mul_rates
tools/prof_kt.sh quick_c2 --steps 12 --warmup 2 | grep -E "seg_sum|rank_seg|sketch_wave|scan_lean|transpose"
