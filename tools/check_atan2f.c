/* tools/check_atan2f.c -- the restatement of glibc 2.35's atan2f / atanf (sysdeps/ieee754/flt-32/{e_atan2f,s_atanf}.c, fdlibm's float code) that the
 * DVB-S2 frame PLL kernel carries (satdump_amd/csrc/dvbs2_demap.hip: s2_atan2f / s2_atanf), compiled for the host and compared with the host libm's
 * atan2f on 6e7 arguments: arbitrary bit patterns, signal-like ranges, tiny ratios, zeros. Expected output: "n=... bad=0".
 *   gcc -O2 -ffp-contract=off -o /tmp/check_atan2f tools/check_atan2f.c -lm && /tmp/check_atan2f
 * (tests/test_dvbs2_pll_math_cpu.py runs the kernel's own copy through the host twin against the same libm.) */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#define MA(a,b,c) ((a)*(b)+(c))
static inline uint32_t fb(float f){uint32_t u;memcpy(&u,&f,4);return u;}
static inline float bf(uint32_t u){float f;memcpy(&f,&u,4);return f;}
static const float atanhi[]={4.6364760399e-01f,7.8539812565e-01f,9.8279368877e-01f,1.5707962513e+00f};
static const float atanlo[]={5.0121582440e-09f,3.7748947079e-08f,3.4473217170e-08f,7.5497894159e-08f};
static const float aT[]={3.3333334327e-01f,-2.0000000298e-01f,1.4285714924e-01f,-1.1111110449e-01f,9.0908870101e-02f,-7.6918758452e-02f,6.6610731184e-02f,-5.8335702866e-02f,4.9768779427e-02f,-3.6531571299e-02f,1.6285819933e-02f};
static float my_atanf(float x){
  float w,s1,s2,z; int32_t ix,hx,id; hx=(int32_t)fb(x); ix=hx&0x7fffffff;
  if(ix>=0x4c000000){ if(ix>0x7f800000) return x+x; if(hx>0) return atanhi[3]+atanlo[3]; else return -atanhi[3]-atanlo[3]; }
  if(ix<0x3ee00000){ if(ix<0x31000000){ if(1.0e30f+x>1.0f) return x; } id=-1; }
  else { x=fabsf(x);
    if(ix<0x3f980000){ if(ix<0x3f300000){ id=0; x=(2.0f*x-1.0f)/(2.0f+x);} else {id=1; x=(x-1.0f)/(x+1.0f);} }
    else { if(ix<0x401c0000){ id=2; x=(x-1.5f)/(1.0f+1.5f*x);} else {id=3; x=-1.0f/x;} } }
  z=x*x; w=z*z;
  s1=z*(aT[0]+w*(aT[2]+w*(aT[4]+w*(aT[6]+w*(aT[8]+w*aT[10])))));
  s2=w*(aT[1]+w*(aT[3]+w*(aT[5]+w*(aT[7]+w*aT[9]))));
  if(id<0) return x-x*(s1+s2);
  else { z=atanhi[id]-((x*(s1+s2)-atanlo[id])-x); return (hx<0)?-z:z; }
}
static float my_atan2f(float y,float x){
  const float tiny=1.0e-30f, pi_o_4=7.8539818525e-01f, pi_o_2=1.5707963705e+00f, pi=3.1415927410e+00f, pi_lo=-8.7422776573e-08f;
  float z; int32_t k,m,hx,hy,ix,iy; hx=(int32_t)fb(x); ix=hx&0x7fffffff; hy=(int32_t)fb(y); iy=hy&0x7fffffff;
  if(ix>0x7f800000||iy>0x7f800000) return x+y;
  if(hx==0x3f800000) return my_atanf(y);
  m=((hy>>31)&1)|((hx>>30)&2);
  if(iy==0){ switch(m){ case 0: case 1: return y; case 2: return pi+tiny; default: return -pi-tiny; } }
  if(ix==0) return (hy<0)? -pi_o_2-tiny: pi_o_2+tiny;
  if(ix==0x7f800000){ if(iy==0x7f800000){ switch(m){ case 0: return pi_o_4+tiny; case 1: return -pi_o_4-tiny; case 2: return 3.0f*pi_o_4+tiny; default: return -3.0f*pi_o_4-tiny; } } else { switch(m){ case 0: return 0.0f; case 1: return -0.0f; case 2: return pi+tiny; default: return -pi-tiny; } } }
  if(iy==0x7f800000) return (hy<0)? -pi_o_2-tiny: pi_o_2+tiny;
  k=(iy-ix)>>23;
  if(k>60) z=pi_o_2+0.5f*pi_lo; else if(hx<0&&k<-60) z=0.0f; else z=my_atanf(fabsf(y/x));
  switch(m){ case 0: return z; case 1: return bf(fb(z)^0x80000000u); case 2: return pi-(z-pi_lo); default: return (z-pi_lo)-pi; }
}
int main(){ uint64_t s=88172645463325252ull; long bad=0,n=0;
  for(long i=0;i<60000000;i++){ s^=s<<13; s^=s>>7; s^=s<<17; float x,y;
    if(i&1){ x=bf((uint32_t)s); y=bf((uint32_t)(s>>32)); }              /* any bit patterns */
    else { x=((int32_t)(uint32_t)s)*(1.0f/1073741824.0f); y=((int32_t)(uint32_t)(s>>32))*(1.0f/1073741824.0f); if((i&6)==2) y*=1e-4f; if((i&6)==4) x=0.0f; } /* signal-like ranges */
    if(x!=x||y!=y) continue; float a=atan2f(y,x), b=my_atan2f(y,x); n++; if(fb(a)!=fb(b) && !(a!=a&&b!=b)){ if(bad<5) printf("y=%a x=%a libm=%a mine=%a\n",y,x,a,b); bad++; } }
  printf("n=%ld bad=%ld\n",n,bad); return 0; }
