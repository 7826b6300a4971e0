#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
static inline float u2f(uint32_t u){float f;memcpy(&f,&u,4);return f;}
static inline uint32_t f2u(float f){uint32_t u;memcpy(&u,&f,4);return u;}
static inline float div1(float a,float d,float y){float q=a*y; float r=__builtin_fmaf(-q,d,a); return __builtin_fmaf(r,y,q);}
static inline float div2(float a,float d,float y){float q=a*y; float r=__builtin_fmaf(-q,d,a); q=__builtin_fmaf(r,y,q); r=__builtin_fmaf(-q,d,a); return __builtin_fmaf(r,y,q);}
static uint64_t s=88172645463325252ull;
static inline uint64_t rnd(){s^=s<<13;s^=s>>7;s^=s<<17;return s;}
int main(int argc,char**argv){
  long N = argc>1? atol(argv[1]) : 400000000L;
  long bad1=0,bad2=0; 
  for(long i=0;i<N;i++){
    uint64_t r=rnd();
    // exponents: d in 2^[-60,60], a in 2^[-40,60]
    uint32_t md=(uint32_t)r&0x7fffff, ma=(uint32_t)(r>>23)&0x7fffff;
    int ed=(int)((r>>46)%121)-60, ea=(int)((r>>53)%101)-40;
    int kind = (r>>60)&7;
    if(kind==0) md=0x7fffff; if(kind==1) md=0; if(kind==2) ma=0x7fffff; if(kind==3) ma=0; if(kind==4){md=0x7ffffe;}
    float d=u2f(((uint32_t)(ed+127)<<23)|md), a=u2f(((uint32_t)(ea+127)<<23)|ma);
    if(r&(1ull<<63)) d=-d; if(rnd()&1) a=-a;
    float y=1.0f/d; float want=a/d;
    float g1=div1(a,d,y), g2=div2(a,d,y);
    if(f2u(g1)!=f2u(want)) { if(bad1<3) printf("div1 mismatch a=%a d=%a got %a want %a\n",a,d,g1,want); bad1++; }
    if(f2u(g2)!=f2u(want)) { if(bad2<3) printf("div2 mismatch a=%a d=%a got %a want %a\n",a,d,g2,want); bad2++; }
  }
  // adversarial: a = q*d rounded neighbours -> quotients near representable / midpoints
  long bad1b=0,bad2b=0, M=N/4;
  for(long i=0;i<M;i++){
    uint64_t r=rnd();
    float d=u2f(((uint32_t)(127+(int)(r%21)-10)<<23)|((uint32_t)(r>>8)&0x7fffff));
    float q=u2f(((uint32_t)(127+(int)((r>>32)%21)-10)<<23)|((uint32_t)(r>>40)&0x7fffff));
    double mid = (double)q + 0.5*(double)(nextafterf(q,INFINITY)-q);   // midpoint
    float a=(float)(mid*(double)d);  // a/d lands near a midpoint
    int k=(int)(rnd()%5)-2; for(int j=0;j<abs(k);j++) a=nextafterf(a, k>0?INFINITY:-INFINITY);
    float y=1.0f/d, want=a/d;
    if(f2u(div1(a,d,y))!=f2u(want)) { if(bad1b<3) printf("adv div1 mismatch a=%a d=%a\n",a,d); bad1b++; }
    if(f2u(div2(a,d,y))!=f2u(want)) { if(bad2b<3) printf("adv div2 mismatch a=%a d=%a\n",a,d); bad2b++; }
  }
  printf("random N=%ld: div1 mismatches %ld, div2 mismatches %ld | near-midpoint M=%ld: div1 %ld div2 %ld\n",N,bad1,bad2,M,bad1b,bad2b);
}
