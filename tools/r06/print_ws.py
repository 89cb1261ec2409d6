import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from molly_amd import ops
print("attn_bwd_workspace (B,T) (2,1024) (2,2048) (16,2048):", ops.attn_bwd_workspace(2, 1024, 16, 8, 128), ops.attn_bwd_workspace(2, 2048, 16, 8, 128),
      ops.attn_bwd_workspace(16, 2048, 16, 8, 128))
