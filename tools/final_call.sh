O=gpurun_out/r5y; mkdir -p $O
(hostname; cat /proc/sys/kernel/random/boot_id; rocm-smi --showuniqueid --showserial 2>/dev/null | grep -i "unique\|serial"; uptime; cat /sys/class/drm/card*/device/vbios_version 2>/dev/null | head -2; cat /sys/module/amdgpu/version 2>/dev/null) > $O/box.txt 2>&1
(timeout 400 python -X faulthandler tools/repro_abort.py 80 2>&1 | tail -6; echo rc ${PIPESTATUS[0]}) > $O/repro.txt
cat $O/box.txt; cat $O/repro.txt
bash tools/evidence_run.sh r5y
