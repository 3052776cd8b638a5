#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define BT_LV 10               // levels of a neighborhood sum: max row count < 1024

// 16-byte vector view of the u16 permutation rows: must be may_alias, the rows are read back
// as unsigned short (without it TBAA lets the compiler move those reads across the refills)
typedef uint4 __attribute__((may_alias)) uint4_alias;

__device__ __forceinline__ void csa32(uint32_t &carry, uint32_t &sum, uint32_t a, uint32_t b, uint32_t c) {
    const uint32_t u = a ^ b;
    carry = (u & c) | (~u & a);            // majority(a,b,c) as one v_bfi_b32
    sum = u ^ c;
}

// adds eight one-bit-per-attribute words into the vertical counter s[0..BT_LV)
__device__ __forceinline__ void vadd8(uint32_t (&s)[BT_LV], const uint32_t (&x)[8]) {
    uint32_t t2a, t2b, t4a, t4b, t8;
    csa32(t2a, s[0], s[0], x[0], x[1]);
    csa32(t2b, s[0], s[0], x[2], x[3]);
    csa32(t4a, s[1], s[1], t2a, t2b);
    csa32(t2a, s[0], s[0], x[4], x[5]);
    csa32(t2b, s[0], s[0], x[6], x[7]);
    csa32(t4b, s[1], s[1], t2a, t2b);
    csa32(t8, s[2], s[2], t4a, t4b);
#pragma unroll
    for (int l = 3; l < BT_LV; ++l) {      // ripple the eights
        const uint32_t c = s[l] & t8;
        s[l] ^= t8;
        t8 = c;
    }
}

template <bool IDENT>
__device__ __forceinline__ void bits_accumulate(const int32_t *__restrict__ cols, int wdt,
                                                const unsigned short *__restrict__ cur, const uint2 *__restrict__ T,
                                                uint32_t (&s0)[BT_LV], uint32_t (&s1)[BT_LV]) {
#pragma unroll
    for (int l = 0; l < BT_LV; ++l) s0[l] = s1[l] = 0;
    for (int t0 = 0; t0 < wdt; t0 += 8) {
        int32_t c[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) c[u] = cols[(t0 + u) * 64];
        uint32_t r[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) r[u] = IDENT ? static_cast<uint32_t>(c[u]) : static_cast<uint32_t>(cur[c[u]]);
        uint32_t x0[8], x1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint2 w = T[r[u]];
            x0[u] = w.x;
            x1[u] = w.y;
        }
        vadd8(s0, x0);
        vadd8(s1, x1);
    }
}

// bit-sliced counter c[0..CL) += mask, with the carries out of the low three levels parked
// in `pend` (a position wraps at most once per 8 increments) and rippled every 8th call
template <int CL>
__device__ __forceinline__ void vcount(uint32_t (&c)[CL], uint32_t &pend, uint32_t m) {
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        const uint32_t k = c[l] & m;
        c[l] ^= m;
        m = k;
    }
    pend |= m;
}

template <int CL>
__device__ __forceinline__ void vflush(uint32_t (&c)[CL], uint32_t &pend) {
    uint32_t m = pend;
#pragma unroll
    for (int l = 3; l < CL; ++l) {
        const uint32_t k = c[l] & m;
        c[l] ^= m;
        m = k;
    }
    pend = 0;
}

template <int LEVELS>
__device__ __forceinline__ unsigned int vextract(const uint32_t (&c)[LEVELS], int bit) {
    unsigned int v = 0;
#pragma unroll
    for (int l = 0; l < LEVELS; ++l) v |= ((c[l] >> bit) & 1u) << l;
    return v;
}


__device__ unsigned lcg(unsigned &st){ st = st*1664525u+1013904223u; return st>>8; }
__global__ void test(int *bad, int *info){
  unsigned st=12345+threadIdx.x;
  const int CL=10;
  int nb=0;
  __shared__ uint2 Ts[64][51];
  __shared__ unsigned short curs[64][51];
  __shared__ int colss[48*64];
  uint2 *T = Ts[threadIdx.x]; unsigned short* cur = curs[threadIdx.x];
  for (int trial=0; trial<50; ++trial){
    int wdt = 8*(1+lcg(st)%6);
    int P = 1+lcg(st)%40;
    for(int r=0;r<50;r++){T[r].x=lcg(st)*977u; T[r].y=lcg(st)*7919u; if(lcg(st)%3) {T[r].x&=lcg(st)*3; T[r].y&=lcg(st)*3;}} T[50].x=T[50].y=0;
    int *cols = colss + threadIdx.x;
    for(int t=0;t<wdt;t++) cols[t*64] = (t < wdt-3) ? lcg(st)%50 : 50;
    uint32_t o0[BT_LV], o1[BT_LV];
    bits_accumulate<true>(cols, wdt, nullptr, T, o0, o1);
    int So[64]; for(int b=0;b<64;b++) So[b]=0; for(int t=0;t<wdt;t++){ uint2 w=T[cols[t*64]]; for(int b=0;b<32;b++){So[b]+=(w.x>>b)&1; So[32+b]+=(w.y>>b)&1;} }
    uint32_t g0[CL],g1[CL],l0[CL],l1[CL]; for(int l=0;l<CL;l++) g0[l]=g1[l]=l0[l]=l1[l]=0; uint32_t gp0=0,gp1=0,lp0=0,lp1=0;
    int G[64], L[64]; for(int b=0;b<64;b++){G[b]=0;L[b]=0;}
    for(int p=0;p<P;p++){
      for(int i=0;i<50;i++) cur[i]=i; for(int i=49;i>0;i--){int j=lcg(st)%(i+1); unsigned short t=cur[i];cur[i]=cur[j];cur[j]=t;} cur[50]=50;
      uint32_t s0[BT_LV], s1[BT_LV];
      bits_accumulate<false>(cols, wdt, cur, T, s0, s1);
      int S[64]; for(int b=0;b<64;b++) S[b]=0; for(int t=0;t<wdt;t++){ uint2 w=T[cur[cols[t*64]]]; for(int b=0;b<32;b++){S[b]+=(w.x>>b)&1; S[32+b]+=(w.y>>b)&1;} }
      for(int b=0;b<64;b++){G[b]+=S[b]>So[b]; L[b]+=S[b]<So[b];}
      uint32_t eq0 = 0xFFFFFFFFu, eq1 = 0xFFFFFFFFu, gt0 = 0, gt1 = 0, lt0 = 0, lt1 = 0;
#pragma unroll
      for (int l = BT_LV - 1; l >= 0; --l) {
          const uint32_t d0 = s0[l] ^ o0[l], d1 = s1[l] ^ o1[l];
          const uint32_t t0 = eq0 & d0, t1 = eq1 & d1;
          gt0 |= t0 & s0[l]; gt1 |= t1 & s1[l]; lt0 |= t0 & o0[l]; lt1 |= t1 & o1[l]; eq0 ^= t0; eq1 ^= t1;
      }
      vcount<CL>(g0, gp0, gt0); vcount<CL>(g1, gp1, gt1); vcount<CL>(l0, lp0, lt0); vcount<CL>(l1, lp1, lt1);
      if ((p & 7) == 7) { vflush<CL>(g0, gp0); vflush<CL>(g1, gp1); vflush<CL>(l0, lp0); vflush<CL>(l1, lp1); }
    }
    vflush<CL>(g0, gp0); vflush<CL>(g1, gp1); vflush<CL>(l0, lp0); vflush<CL>(l1, lp1);
    for(int b=0;b<32;b++){ if (vextract<CL>(g0,b)!=(unsigned)G[b]||vextract<CL>(g1,b)!=(unsigned)G[32+b]||vextract<CL>(l0,b)!=(unsigned)L[b]||vextract<CL>(l1,b)!=(unsigned)L[32+b]){nb++; if (threadIdx.x==0 && nb==1){info[0]=trial; info[1]=b; info[2]=vextract<CL>(g0,b); info[3]=G[b]; info[4]=So[b];} break;} }
  }
  atomicAdd(bad, nb);
}
int main(){ int *d; hipMalloc(&d, 64); hipMemset(d,0,64); test<<<1,64>>>(d, d+1); int h[16]; hipMemcpy(h,d,64,hipMemcpyDeviceToHost); printf("bad=%d info %d %d %d %d %d\n",h[0],h[1],h[2],h[3],h[4],h[5]); }
