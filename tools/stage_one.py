import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llckbdm_amd import datasets
from llckbdm_amd.engine import Engine
eng=Engine(0)
sigs,_,_=datasets.config2(0)
ms=np.array([int(x) for x in sys.argv[1:]] or [400],dtype=np.int32)
plan=eng.plan(1,sigs.shape[1],np.zeros(len(ms),np.int32),ms,ms,p=1,q=0.0,dwell=5e-4)
plan.upload(sigs); plan.execute(); plan.execute()
st=plan.stage_ms()
print(' '.join('%s=%.1f'%(k.replace('k_',''),v) for k,v in st.items() if v>0.5), 'total %.1f'%sum(st.values()))
