import numpy as np, sys
def icosphere(sub):
    t=(1+5**0.5)/2
    v=[(-1,t,0),(1,t,0),(-1,-t,0),(1,-t,0),(0,-1,t),(0,1,t),(0,-1,-t),(0,1,-t),(t,0,-1),(t,0,1),(-t,0,-1),(-t,0,1)]
    f=[(0,11,5),(0,5,1),(0,1,7),(0,7,10),(0,10,11),(1,5,9),(5,11,4),(11,10,2),(10,7,6),(7,1,8),(3,9,4),(3,4,2),(3,2,6),(3,6,8),(3,8,9),(4,9,5),(2,4,11),(6,2,10),(8,6,7),(9,8,1)]
    v=[np.array(p,float)/np.linalg.norm(p) for p in v]
    for _ in range(sub):
        cache={}; nf=[]
        def mid(a,b):
            k=(min(a,b),max(a,b))
            if k not in cache:
                m=(v[a]+v[b]); m/=np.linalg.norm(m); v.append(m); cache[k]=len(v)-1
            return cache[k]
        for a,b,c in f:
            ab,bc,ca=mid(a,b),mid(b,c),mid(c,a)
            nf+=[(a,ab,ca),(b,bc,ab),(c,ca,bc),(ab,bc,ca)]
        f=nf
    return np.array(v),f
sub=int(sys.argv[1]); out=sys.argv[2]
v,f=icosphere(sub)
with open(out,'w') as o:
    o.write("o ico\n")
    for p in v: o.write("v %.6f %.6f %.6f\n"%tuple(p*120+np.array([0,0,250])))   # scene offset -500 in z is added by open_obj
    for a,b,c in f: o.write("f %d %d %d\n"%(a+1,b+1,c+1))
print(len(v),len(f))
