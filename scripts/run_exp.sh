mkdir -p gpurun_out
O=gpurun_out/exp16.txt
: > $O
export BMX_LIB_NAME=libbmx_diag.so
python scripts/kexp.py --config 5 --windows 131072 --tag c5_lds >> $O 2>&1
BMX_NO_LDS=1 python scripts/kexp.py --config 5 --windows 131072 --tag c5_l2 >> $O 2>&1
BMX_NO_LDS=1 python scripts/kexp.py --config 3 --windows 131072 --tag c3_l2 >> $O 2>&1
BMX_NO_LDS=1 BMX_MOM_SLOTS=64 python scripts/kexp.py --config 5 --windows 131072 --tag c5_l2_64slots >> $O 2>&1
cat $O | cut -c1-130
