nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; python3 -c "import os; print('affinity', len(os.sched_getaffinity(0)), 'cpu_count', os.cpu_count())"; free -g | head -2; df -h /dev/shm | tail -1
