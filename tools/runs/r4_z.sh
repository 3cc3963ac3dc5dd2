mkdir -p gpurun_out/r4z
( timeout -k 10 500 python3 tools/ddp_check.py --ranks 2 && timeout -k 10 500 python3 tools/ddp_check.py --ranks 4 ) > gpurun_out/r4z/ddp_check.txt 2>&1
rc=$?; grep -E "ddp_check|rank [0-9]:" gpurun_out/r4z/ddp_check.txt | cut -c1-160
exit $rc
