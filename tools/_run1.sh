set -e
export TMPDIR=/tmp
O=gpurun_out/r5c; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
AB_ROUNDS=5 timeout -k 10 200 python3 tools/ab_k1_graph.py build/variants/libso3proj_r03.so build/variants/libso3proj_r04.so build/variants/libso3proj_r05c.so > $O/ab_k1_graph.txt 2>&1
cat $O/ab_k1_graph.txt
AB_ROUNDS=4 timeout -k 10 300 python3 tools/ab_v2.py build/variants/libso3proj_r04.so build/variants/libso3proj_r05c.so > $O/ab_v2.txt 2>&1
cat $O/ab_v2.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $O/sq1 -- python3 tools/k1_loop.py 20 > /dev/null 2> $O/sq1.log
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CU_CYCLES --output-format csv -d $O/sq2 -- python3 tools/k1_loop.py 20 > /dev/null 2> $O/sq2.log
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $O/sq3 -- python3 tools/k1_loop.py 20 > /dev/null 2> $O/sq3.log
python3 tools/summarize_sq.py $O r05c_tmp
