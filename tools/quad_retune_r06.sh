#!/bin/bash
# round 6: the quad path's shape table re-measured with two sets of masks in flight (tools/quad_probe.py per width; the library's own choice is the first line of each block)
P="python tools/quad_probe.py --shapes"
$P 4,8,12:4,8,8:4,12,8:4,16,12:8,16,12:8,12,12:8,8,8:8,4,8:16,4,8:8,6,8:4,6,8:8,20,12 2048 512 2048 1024 2048 2048 2048 4096 2048 8192 2048 16384 2>&1 | grep -v "amdgpu.ids"
$P 4,12,16:4,8,12:4,8,16:8,8,16:8,4,12:8,4,16:4,12,12:2,12,12:4,6,12:8,6,12:4,4,12:8,4,8 4096 1024 4096 2048 4096 4096 4096 8192 4096 16384 2>&1 | grep -v "amdgpu.ids"
$P 2,8,16:4,8,16:8,6,16:4,4,12:4,6,16:8,4,16:4,6,12:4,4,16:2,6,12 6144 1024 6144 2048 6144 6144 2>&1 | grep -v "amdgpu.ids"
$P 2,8,16:4,8,16:4,4,16:2,6,16:4,6,16:2,4,16:4,4,12 8192 512 8192 1024 8192 2048 10240 1024 12288 768 16384 512 2>&1 | grep -v "amdgpu.ids"
