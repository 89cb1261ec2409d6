O=gpurun_out/r04; mkdir -p $O
{ python tools/r04/attn_stamp.py; MOLLY_ATTN_LDS_PAD=40960 python tools/r04/attn_stamp.py; } 2>&1 | grep -v amdgpu.ids | tee $O/attn_stamp.log
