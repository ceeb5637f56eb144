python tools/ab_lib.py run w4 nt --rounds 2 --only c5,c5big > gpurun_out/r3_t14_ab.log 2>&1
tail -4 gpurun_out/r3_t14_ab.log
python tools/ab_lib.py run r2 tree --rounds 2 --only c3,c2roll,c5,c4share,central10x5,central32x10,c3rf > gpurun_out/r3_t14_ab2.log 2>&1
tail -9 gpurun_out/r3_t14_ab2.log
