mkdir -p gpurun_out
python scripts/e2e_timing.py > gpurun_out/e2e_r02.txt 2>&1; grep "CLI end" gpurun_out/e2e_r02.txt
for i in 1 2 3; do s=$(date +%s.%N); BMX_TRACE=1 python BalLeRMixPlus_amd.py -i /tmp/bmx_e2e_in.txt --spect /tmp/bmx_e2e_spect.txt -o /tmp/o.txt 2>&1 | grep -E "bmx cli" | tr '\n' ';'; e=$(date +%s.%N); echo " wall $(echo "$e - $s" | bc)"; done
s=$(date +%s.%N); python -c "import numpy" ; e=$(date +%s.%N); echo "python+numpy $(echo "$e - $s" | bc)"
