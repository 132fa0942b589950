#!/bin/bash
# What the GPU box offers besides the GPU: host cores, memory (and its cgroup limit), scratch space.  Usage: scripts/box_probe.sh > gpurun_out/box.txt
echo "nproc: $(nproc)"; lscpu | grep -E "Model name|Socket|Core|Thread|NUMA node\(s\)"
free -g | head -3
for f in /sys/fs/cgroup/memory.max /sys/fs/cgroup/memory/memory.limit_in_bytes /sys/fs/cgroup/cpu.max; do [ -f $f ] && echo "$f: $(cat $f)"; done
df -h /tmp /dev/shm . 2>/dev/null
ulimit -a | grep -E "max memory|virtual|open files|locked"
rocm-smi --showmeminfo vram 2>/dev/null | grep -i total | head -2
