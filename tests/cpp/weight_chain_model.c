/* CPU model of weight_stretches() (voxblox_amd/csrc/vbx_kernels_tsdf.hpp): the weight chain
 *     W <- min(max_weight, W + w_j)        (updateTsdfVoxel, tsdf_integrator.cc:188-208)
 * of 64 updates evaluated in stretches — inside one binade [2^e, 2^(e+1)) the weight is an integer
 * multiple k of u = 2^(e-23) and fl(W + w) = (k + rn(w / u)) * u, so a stretch is an integer prefix sum —
 * against the plain sequential float loop, bit for bit, over random weights (1/z^2, constants, powers
 * of two that produce ties, zeros, negative values) and starting weights (zero, small, binade
 * boundaries, near max_weight).  TEST INFRASTRUCTURE ONLY (tests/test_oracle_known_answers.py runs it);
 * on the device every result of the chain is verified against the literal update as well. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static uint32_t fb(float f){uint32_t u;memcpy(&u,&f,4);return u;}
int main(){
  srand(7);
  const float maxW=10000.f;
  long stretches=0, literal=0, chunks=0;
  for(int trial=0;trial<400000;++trial){
    float uw[64]; int cnt = (rand()%8==0)? 1+rand()%64 : 64;
    int mode=rand()%6;
    float W;
    switch(rand()%5){case 0:W=0;break;case 1:W=(float)rand()/RAND_MAX*20;break;case 2:W=(float)rand()/RAND_MAX*12000; if(W>maxW)W=maxW;break;case 3:W=ldexpf(1.f,rand()%14)-((rand()%3)?0.f:0.37f);break;default:W=9990+(float)rand()/RAND_MAX*10;}
    for(int j=0;j<64;++j){
      float z=0.5f+(float)rand()/RAND_MAX*6;
      switch(mode){case 0:uw[j]=1.f;break;case 1:uw[j]=1.f/(z*z);break;case 2:uw[j]=ldexpf(1.f,-(rand()%30));break;case 3:uw[j]=(rand()%4==0)?0.f:1.f/(z*z);break;case 4:uw[j]=(float)(rand()%7)*0.5f*ldexpf(1.f,-(rand()%14));break;default:uw[j]=(rand()%50==0)?-0.01f:0.3f/(z*z);}
      if(j>=cnt)uw[j]=0;
    }
    // sequential
    float Ws=W, Wm_ref[64];
    for(int j=0;j<cnt;++j){Wm_ref[j]=Ws; float nw=Ws+uw[j]; if(!(nw<1e-6f)) Ws=fminf(maxW,nw);}
    // stretches
    float Wrun=W, Wm[64]; int start=0; ++chunks;
    while(start<cnt){
      int allpos=1; for(int j=0;j<64;++j) if(!(uw[j]>=0)) allpos=0;
      if(Wrun==maxW&&allpos){for(int j=start;j<64;++j)Wm[j]=Wrun;break;}
      uint32_t wb=fb(Wrun); int e=(int)((wb>>23)&0xFF)-127; int k0=(int)(wb&0x7FFFFF)|0x800000;
      int pre=0, stop=64; float Wafter[64]; int prej[64], incj[64];
      for(int j=0;j<64;++j){
        int active=j>=start&&j<cnt;
        float f=active?ldexpf(uw[j],23-e):0.f;
        int exact=Wrun>=1e-6f&&uw[j]>=0&&f<8388608.f&&(f-floorf(f))!=0.5f;
        int inc=(active&&exact)?(int)rintf(f):0;
        pre+=inc; prej[j]=pre; incj[j]=inc;
        Wafter[j]=ldexpf((float)(k0+pre),e-23);
        int good=!active||(exact&&(k0+pre)<=0xFFFFFF&&Wafter[j]<=maxW);
        if(!good&&stop==64)stop=j;
      }
      if(stop>start){for(int j=start;j<stop;++j)Wm[j]=ldexpf((float)(k0+prej[j]-incj[j]),e-23);Wrun=Wafter[stop-1];start=stop;++stretches;}
      else{Wm[start]=Wrun;float nw=Wrun+uw[start];if(!(nw<1e-6f))Wrun=fminf(maxW,nw);++start;++literal;}
    }
    if(fb(Wrun)!=fb(Ws)){printf("MISMATCH final trial %d mode %d W %g: %g vs %g\n",trial,mode,W,Wrun,Ws);return 1;}
    for(int j=0;j<cnt;++j)if(fb(Wm[j])!=fb(Wm_ref[j])){printf("MISMATCH lane %d trial %d mode %d\n",j,trial,mode);return 1;}
  }
  printf("ok chunks %ld stretches %ld literal %ld\n",chunks,stretches,literal);
  return 0;
}
