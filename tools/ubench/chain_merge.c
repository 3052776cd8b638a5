// Does the masked-rejection chain of np.random.permutation forget its past?  If a run of the rule started at an ARBITRARY word
// of the MT19937 stream (fresh state i = k - 1) fell back onto the true chain's permutation boundaries after a warm-up of W
// permutations, the stream could be cut into blocks and drawn in parallel (each block: W warm-up permutations, then its own).
// Measured: it does not -- a run started off a boundary stays off (the lag between two runs is conserved: within a level the
// gap in accepts halves, at every level boundary it doubles again).  Failures per 20000 random starts, W = 1..4:
//   k = 3789: 19933 19873 19817 19800      k = 20000: 19987 19970 19957 19950      k = 100: 18652 17839 17223 16580
// So the chain (which word each permutation starts at) is sequential in the whole stream; everything else -- which draws are
// accepted inside a permutation, the swaps, the composition of the row maps -- is independent per permutation once the
// start offsets are known.  build: gcc -O2 chain_merge.c -o chain_merge ; run: ./chain_merge k P trials
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
static uint32_t mt[624]; static int mti;
static void seed(uint32_t s){ mt[0]=s; for(int i=1;i<624;i++) mt[i]=1812433253u*(mt[i-1]^(mt[i-1]>>30))+i; mti=624; }
static uint32_t next(){ if(mti>=624){ for(int i=0;i<624;i++){ uint32_t y=(mt[i]&0x80000000u)|(mt[(i+1)%624]&0x7fffffffu); mt[i]=mt[(i+397)%624]^(y>>1)^((y&1)?0x9908b0dfu:0);} mti=0;} uint32_t y=mt[mti++]; y^=y>>11; y^=(y<<7)&0x9d2c5680u; y^=(y<<15)&0xefc60000u; y^=y>>18; return y; }
static uint32_t *w; static size_t NW;
// run one permutation of k items from offset o; returns end offset
static size_t run(size_t o, int k){ for(int i=k-1;i>=1;--i){ uint32_t m=i; m|=m>>1;m|=m>>2;m|=m>>4;m|=m>>8;m|=m>>16; while(1){ if(o>=NW) return NW; uint32_t v=w[o++]&m; if(v<=(uint32_t)i) break; } } return o; }
int main(int argc,char**argv){ int k=atoi(argv[1]); int P=atoi(argv[2]); int trials=atoi(argv[3]);
  // expected words per perm
  double mu=0; for(int i=k-1;i>=1;--i){ uint32_t m=i; m|=m>>1;m|=m>>2;m|=m>>4;m|=m>>8;m|=m>>16; mu+=(double)(m+1)/(i+1);} 
  NW=(size_t)(mu*(P+8))+100000; w=malloc(NW*4); seed(12345); for(size_t i=0;i<NW;i++) w[i]=next();
  uint8_t *isb=calloc(NW+1,1); size_t o=0; isb[0]=1; int np=0; while(o<NW && np<P+4){ o=run(o,k); if(o<NW) isb[o]=1; np++; }
  size_t lim=o; printf("k=%d mu=%.1f words/perm, true perms %d, words %zu\n",k,mu,np,lim);
  srand(1); for(int W=1;W<=4;W++){ int fail=0; for(int t=0;t<trials;t++){ size_t s=(size_t)(((double)rand()/RAND_MAX)*(lim-(W+1)*mu*1.5)); size_t e=s; for(int r=0;r<W;r++) e=run(e,k); if(e>=lim) {t--; continue;} if(!isb[e]) fail++; } printf("  W=%d fail %d / %d\n",W,fail,trials);} return 0; }
