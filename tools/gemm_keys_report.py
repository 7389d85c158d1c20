import sys
sys.path.insert(0,'tools'); sys.path.insert(0,'.')
import torch, gemm_tune as G
dev=torch.device('cuda:0')
name=sys.argv[1]
step,tokens=G.build(name,dev)
for _ in range(5): step()
ms=G.timed_steps(step,10)
keys=G.per_key_us(step,5)
tot=0
rows=[]
for k,(us,cnt) in keys.items():
    op,M,N,K,epi,acc=k
    fl=2.0*M*N*K
    rows.append((us*cnt, G.OPN[op],M,N,K,epi,acc,us,cnt,fl/us/1e6))
    tot+=us*cnt
rows.sort(reverse=True)
print("%s: %.3f ms/step; GEMMs %.3f ms"%(name,ms,tot/1e3))
for r in rows:
    print("  %s %5dx%5dx%5d epi%d acc%d: %7.1f us x %4.1f = %7.1f us/step  %5.1f TF (%.2f)"%(r[1],r[2],r[3],r[4],r[5],r[6],r[7],r[8],r[0],r[9],r[9]/157.3))
