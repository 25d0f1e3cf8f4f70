// Experiment (not product code): are workgroup-scope global atomics on per-XCD private counters executed in the
// XCD's L2 (fast) and still exact?  Compared with default agent-scope atomics on one shared array.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
__device__ __forceinline__ uint32_t xcc_id(){ return __builtin_amdgcn_s_getreg((3<<11)|20) & 7u; }
__device__ __forceinline__ uint32_t hashk(uint32_t i){ i*=2654435761u; i^=i>>15; i*=2246822519u; i^=i>>13; return i; }
template<int MODE> __global__ void k(uint32_t* cnt, uint32_t* rank, uint32_t* xcd_of, int n, int nkeys){
  const uint32_t x = xcc_id();
  for (int i = blockIdx.x*blockDim.x+threadIdx.x; i < n; i += gridDim.x*blockDim.x){
    uint32_t key = hashk(i) % nkeys, r;
    if (MODE==0) r = atomicAdd(&cnt[key],1u);
    else if (MODE==1) r = __hip_atomic_fetch_add(&cnt[x*nkeys+key],1u,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_WORKGROUP);
    else r = __hip_atomic_fetch_add(&cnt[x*nkeys+key],1u,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_AGENT);
    rank[i]=r; if (MODE) xcd_of[i]=x;
  }
}
int main(){
  const int n=5000000, nkeys=131072;
  uint32_t *cnt,*rank,*xo; CK(hipMalloc(&cnt,8*nkeys*4)); CK(hipMalloc(&rank,n*4)); CK(hipMalloc(&xo,n*4));
  hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  for (int mode=0; mode<3; ++mode){
    float best=1e9;
    for (int rep=0; rep<5; ++rep){
      CK(hipMemset(cnt,0,8*nkeys*4)); CK(hipDeviceSynchronize());
      hipEventRecord(a);
      if(mode==0) k<0><<<2048,256>>>(cnt,rank,xo,n,nkeys); else if(mode==1) k<1><<<2048,256>>>(cnt,rank,xo,n,nkeys); else k<2><<<2048,256>>>(cnt,rank,xo,n,nkeys);
      hipEventRecord(b); CK(hipDeviceSynchronize()); float ms; hipEventElapsedTime(&ms,a,b); if(ms<best)best=ms;
    }
    std::vector<uint32_t> hc(8*nkeys), hr(n), hx(n);
    CK(hipMemcpy(hc.data(),cnt,8*nkeys*4,hipMemcpyDeviceToHost)); CK(hipMemcpy(hr.data(),rank,n*4,hipMemcpyDeviceToHost)); CK(hipMemcpy(hx.data(),xo,n*4,hipMemcpyDeviceToHost));
    // exactness: every (array slot) must have ranks 0..cnt-1 exactly once
    uint64_t total=0, want=0, got=0; for (auto c: hc){ total+=c; want += (uint64_t)c*(c-1)/2; } for (int i=0;i<n;++i) got+=hr[i];
    std::vector<uint32_t> seen(8*nkeys,0); bool uniq=true;
    // cheap uniqueness proxy: sum of ranks equals sum c(c-1)/2 and max rank < count
    for (int i=0;i<n;++i){ uint32_t key=( (uint32_t)i*2654435761u ); key^=key>>15; key*=2246822519u; key^=key>>13; key%=nkeys; uint32_t slot=(mode?hx[i]*nkeys:0)+key; if(hr[i]>=hc[slot]) uniq=false; }
    uint32_t xs[8]={0}; if(mode) for(int i=0;i<n;++i) xs[hx[i]&7]++;
    printf("mode %d: %.1f us  (%.1f G atomics/s)  total=%llu (want %d) ranksum ok=%d maxrank ok=%d  xcd hist %u %u %u %u %u %u %u %u\n",mode,best*1e3,n/best/1e6,(unsigned long long)total,n,(int)(want==got),(int)uniq,xs[0],xs[1],xs[2],xs[3],xs[4],xs[5],xs[6],xs[7]);
  }
  return 0;
}
