import numpy as np, sys
d=np.load(sys.argv[1]); G=int(sys.argv[2]); NS=int(sys.argv[3]) if len(sys.argv)>3 else 400
bx,by,bz,cx,cy,cz=[d[k].astype(np.int64) for k in ('bx','by','bz','cx','cy','cz')]
rx,ry,rz=int(d['rx']),int(d['ry']),int(d['rz']); n=len(bx)
TY=TZ=8; ryT=(ry+TY-1)//TY
def row_index(y,z): return (((z//TZ)*ryT+(y//TY))*(TY*TZ))+(z%TZ)*TY+(y%TY)
key=bx+rx*row_index(by,bz)
order=np.lexsort((np.arange(n),key)); skey=key[order]
nrows_t=ryT*TY*((rz+TZ-1)//TZ)*TZ
cum=np.zeros(rx*nrows_t+1,dtype=np.int64); np.add.at(cum,skey+1,1); cum=np.cumsum(cum)
X,Y,Z=cx[order],cy[order],cz[order]
def rng(y,z,xlo,xhi):
    if y<0 or y>=ry or z<0 or z>=rz: return 0,0
    xlo,xhi=max(xlo,0),min(xhi,rx-1)
    if xlo>xhi: return 0,0
    b=rx*row_index(y,z); return int(cum[b+xlo]),int(cum[b+xhi+1])
rs=np.random.RandomState(2); ng=n//G
tot=[];kk=[];perstep=[];need=[]
for g in rs.choice(ng,size=NS,replace=False):
    sl=slice(g*G,g*G+G); x,y,z=X[sl],Y[sl],Z[sl]
    rows={}
    for i in range(G):
        r=(int(y[i]),int(z[i])); 
        if r in rows: rows[r][0]=min(rows[r][0],x[i]); rows[r][1]=max(rows[r][1],x[i])
        else: rows[r]=[x[i],x[i]]
    kk.append(len(rows))
    for dz in (-1,0,1):
        for dy in (-1,0,1):
            # dedupe target rows in this step
            used=0; raw=0
            for (yy,zz),(x0,x1) in rows.items():
                A,E=rng(yy+dy,zz+dz,x0-1,x1+1)
                used+=(E-A+63)//64*64; raw+=E-A
            perstep.append(used); need.append(raw)
    # whole-neighbourhood union (all steps at once): distinct rows dilated
    allrows={}
    for (yy,zz),(x0,x1) in rows.items():
        for dz in (-1,0,1):
            for dy in (-1,0,1):
                r=(yy+dy,zz+dz)
                if r in allrows: allrows[r][0]=min(allrows[r][0],x0-1); allrows[r][1]=max(allrows[r][1],x1+1)
                else: allrows[r]=[x0-1,x1+1]
    t=0
    for (yy,zz),(x0,x1) in allrows.items():
        A,E=rng(yy,zz,x0,x1); t+=(E-A+63)//64*64
    tot.append(t)
kk=np.array(kk);perstep=np.array(perstep);tot=np.array(tot);need=np.array(need)
print(f"group of {G} slots: distinct current rows mean {kk.mean():.1f} p95 {np.percentile(kk,95):.0f} max {kk.max()}")
print(f"  per-step staged slots (64-chunks): mean {perstep.mean():.0f} p95 {np.percentile(perstep,95):.0f} p99 {np.percentile(perstep,99):.0f} max {perstep.max()}  (raw {need.mean():.0f}) -> per target per step {perstep.mean()/G:.2f}")
print(f"  whole neighbourhood at once: mean {tot.mean():.0f} p95 {np.percentile(tot,95):.0f} max {tot.max()} slots -> per target {tot.mean()/G:.2f}; x12B = {tot.mean()*12/1024:.0f} KB mean, {np.percentile(tot,95)*12/1024:.0f} KB p95, {tot.max()*12/1024:.0f} KB max")
