#include "draws.h"
#include <cstdio>
#include <vector>
#include <chrono>
int main(){
  const int k=3789,P=1000; DrawStream* ds=draw_stream_new(0);
  std::vector<uint32_t> st((size_t)P*k);
  auto t0=std::chrono::steady_clock::now();
  for(int q=0;q<P;q++) draw_stream_targets(ds,k,st.data()+(size_t)q*k);
  auto t1=std::chrono::steady_clock::now();
  printf("draws: %.2f ms\n", std::chrono::duration<double,std::milli>(t1-t0).count());
  // dump first perm's first/last targets + checksum
  unsigned long long h=0; for(size_t i=0;i<st.size();i++) if ((i%k)<(size_t)(k-1)) h=h*1315423911ull+st[i];
  printf("hash %llu first %u %u %u\n",h,st[0],st[1],st[2]);
}
