#!/bin/bash
lscpu | grep -i "numa\|socket\|thread" | head
for d in /sys/class/drm/card*/device; do echo "$d numa=$(cat $d/numa_node 2>/dev/null) cpus=$(cat $d/local_cpulist 2>/dev/null)"; done 2>/dev/null | head
cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null | head -2
python3 -c "import os; print('affinity', len(os.sched_getaffinity(0)))"
mount | grep shm
