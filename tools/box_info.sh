#!/bin/bash
# What a GPU box says about itself (read-only; no setting is changed): driver / firmware versions, partition modes, clocks, power state.  Companion of
# tools/box_probe.py, which measures whether the next-weight L2 prefetch pays on the box.  Usage: bash tools/box_info.sh > gpurun_out/box_info.txt
echo "== host"; uname -r; cat /sys/module/amdgpu/version 2>/dev/null; hostname | md5sum | cut -c1-8
echo "== rocm-smi"; rocm-smi --showdriverversion --showvbios --showmemorypartition --showcomputepartition --showperflevel --showpower --showmaxpower --showclocks --showtemp 2>&1 | grep -v "^$\|=====" | head -60
echo "== fw"; rocm-smi --showfwinfo 2>&1 | grep -v "^$\|=====" | head -40
echo "== sysfs"
for d in /sys/class/drm/card*/device; do
  for f in current_compute_partition current_memory_partition available_memory_partition mem_busy_percent pp_dpm_sclk pp_dpm_mclk pp_dpm_fclk pp_dpm_socclk power_dpm_force_performance_level vbios_version xgmi_hive_info/xgmi_hive_id mem_info_vram_total mem_info_vram_vendor; do
    [ -r $d/$f ] && echo "$d/$f: $(tr '\n' ' ' < $d/$f)"
  done
done 2>/dev/null | head -60
echo "== rocminfo caches"; rocminfo 2>/dev/null | grep -i -E "Marketing|L2:|L3:|Cacheline|Compute Unit|Max Clock|Chip ID|ASIC Revision|Uuid" | sort | uniq -c | head -30
echo "== kfd topology (gpu node caches)"
for n in /sys/class/kfd/kfd/topology/nodes/*; do
  if grep -q "simd_count [1-9]" $n/properties 2>/dev/null; then
    grep -E "simd_count|array_count|num_xcc|max_engine_clk|fw_version|sdma_fw_version|device_id|unique_id|num_cp_queues|cu_per_simd_array" $n/properties | tr '\n' ' '; echo
    for c in $n/caches/*; do grep -E "level|size|type" $c/properties | tr '\n' ' '; echo; done 2>/dev/null | sort | uniq -c | head -12
  fi
done
