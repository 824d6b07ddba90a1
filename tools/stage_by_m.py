import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llckbdm_amd import datasets
from llckbdm_amd.engine import Engine
eng=Engine(0)
sigs,_,_=datasets.config2(0)
for ms in ([100],[200],[300],[400],[398,400],[400]*8, list(range(100,401,2))):
    ms=np.array(ms,dtype=np.int32)
    plan=eng.plan(1,sigs.shape[1],np.zeros(len(ms),np.int32),ms,ms,p=1,q=0.0,dwell=5e-4)
    plan.upload(sigs); plan.execute(); plan.execute()
    st=plan.stage_ms(); res=plan.download()
    print('m',list(ms[:3]),'n',len(ms),'status',int(res.status.max()),' '.join('%s=%.1f'%(k.replace('k_',''),v) for k,v in st.items() if v>0.5), 'total %.1f'%sum(st.values()))
    plan.close()
