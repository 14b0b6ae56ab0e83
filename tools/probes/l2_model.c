// tools/probes/l2_model.c -- round 5: a cache model of the DIRECT-ADDRESSED short-seed probe structure that the round-4 review
// proposed for the C3 (--sensitive) stream kernel, written BEFORE building the kernel to see what it could buy.
//
// Models ONE XCD's L2 (4 MB, 128-byte lines, 16-way LRU) and the accesses of the key class it owns (class = dense seed code mod 8):
// an i.i.d. reference stream with BLOSUM62 background frequencies reduced to the 11 classes of the seed alphabet, a 10 000 x 300
// query block, shape 1011110111 (weight 8: 11^8 = 2.14e8 codes). Per reference window of the class: one probe of the exact
// membership map; on a join: the rank structure, the dense slot of the seed, the list of query positions, the folded query
// windows -- and, competing for the same L2, the streamed data (class nibbles, window maps, the reference letters around a join).
//   layout 0: 1 bit per code + a 32-bit rank prefix per 64 codes      layout 1: interleaved { 32 bits, 32-bit prefix } per 32 codes
//   layout 2: 1 bit per code + a prefix per 256 codes                 slot bytes: 16 (key, head, flags) or 8 (head, flags)
// usage: l2_model LAYOUT L2_MB REFERENCE_LETTERS SLOT_BYTES NO_STREAM      (build: gcc -O2 -o l2_model l2_model.c -lm)
// Output: accesses and misses per structure, and the misses of all eight classes scaled to one launch (3.0e8 letters) --
// the number to hold against the kernel's measured TCC_EA0_RDREQ (1.78e8 per launch with the hashed structure of round 4).
// Result (profiles/r05_l2_model_direct_addressing.txt): 1.3 - 1.7e8 + 3e7 of streamed lines, i.e. NO fewer than the hashed
// structure: what the exact map saves in slot lines (dense 8-byte slots miss 46 % instead of ~90 %) it spends on its own cold lines
// (12 % of 3.75e7 probes per class). The join's random accesses do not fit 4 MB whatever the layout; the structure was not built.
// L2 model for the direct-addressed short-seed stream: one XCD (class 0), 4 MB, 128-B lines, 16-way LRU.
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
static const double freq[20] = {0.078,0.051,0.045,0.054,0.019,0.043,0.063,0.074,0.022,0.051,0.091,0.057,0.022,0.039,0.052,0.071,0.058,0.013,0.032,0.064}; // ARNDCQEGHILKMFPSTWYV
static const int red[20] = {0,1,2,2,3,2,2,4,5,6,6,1,6,7,8,9,9,7,7,6};
#define R 11
static uint64_t rng=88172645463325252ull; static inline uint64_t xr(){rng^=rng<<13;rng^=rng>>7;rng^=rng<<17;return rng;}
static int draw(){ double u=(xr()>>11)*(1.0/9007199254740992.0),c=0; for(int i=0;i<20;i++){c+=freq[i]; if(u<c) return i;} return 19; }
// cache
#define WAYS 16
static int SETS; static uint64_t *tags; static uint32_t *lru; static uint32_t tick;
static uint64_t hits[8], miss[8];
static void cinit(double mb){ SETS=(int)(mb*1048576/128/WAYS); tags=calloc((size_t)SETS*WAYS,8); lru=calloc((size_t)SETS*WAYS,4); memset(tags,0xff,(size_t)SETS*WAYS*8);} 
static void acc(uint64_t addr,int kind){ uint64_t line=addr>>7; int s=(int)((line*0x9E3779B97F4A7C15ull>>40)%SETS); uint64_t*t=tags+(size_t)s*WAYS; uint32_t*l=lru+(size_t)s*WAYS; ++tick; int v=0; for(int w=0;w<WAYS;w++){ if(t[w]==line){l[w]=tick;hits[kind]++;return;} if(l[w]<l[v])v=w;} t[v]=line;l[v]=tick;miss[kind]++; }
int main(int argc,char**argv){
  int layout=argc>1?atoi(argv[1]):0; double mb=argc>2?atof(argv[2]):4.0; long NT=argc>3?atol(argv[3]):40000000; int slotB=argc>4?atoi(argv[4]):16; int nostream=argc>5?atoi(argv[5]):0;
  const int pos[8]={0,2,3,4,5,7,8,9}; // 1011110111
  long NQ=3000000; 
  uint8_t*q=malloc(NQ+32); for(long i=0;i<NQ+32;i++) q[i]=red[draw()];
  uint32_t*qc=malloc(NQ*4); // dense code
  for(long i=0;i<NQ;i++){ uint32_t d=0; for(int k=0;k<8;k++) d=d*R+q[i+pos[k]]; qc[i]=d; }
  // class 0 only: codes with d&7==0; sort
  long n0=0; uint64_t*keys=malloc(NQ*8); for(long i=0;i<NQ;i++) if((qc[i]&7)==0) keys[n0++]=((uint64_t)(qc[i]>>3)<<32)|(uint64_t)i;
  int cmp(const void*a,const void*b){uint64_t x=*(uint64_t*)a,y=*(uint64_t*)b;return x<y?-1:x>y;} qsort(keys,n0,8,cmp);
  uint32_t NB=214358881/8+1; uint8_t*bm=calloc(NB/8+1,1); uint32_t*rank=malloc((size_t)(NB/32+2)*4); // group idx via map
  // groups
  long ng=0; uint32_t*gstart=malloc(n0*4),*gcount=malloc(n0*4),*gB=malloc(n0*4);
  for(long i=0;i<n0;){ long e=i; uint32_t B=keys[i]>>32; while(e<n0&&(keys[e]>>32)==B)e++; gB[ng]=B;gstart[ng]=i;gcount[ng]=e-i;ng++; bm[B>>3]|=1<<(B&7); i=e; }
  // rank lookup by binary search at sim time (cost irrelevant)
  fprintf(stderr,"class0: %ld positions, %ld groups\n",n0,ng);
  cinit(mb);
  // address spaces
  uint64_t A_BM=1ull<<40,A_PF=2ull<<40,A_SL=3ull<<40,A_QL=4ull<<40,A_FD=5ull<<40,A_TC=6ull<<40,A_TD=7ull<<40;
  uint8_t*t=malloc(64); long joins=0,pairs=0,probes=0; uint8_t win[16]; for(int i=0;i<16;i++)win[i]=red[draw()];
  for(long p=0;p<NT;p++){
    memmove(win,win+1,15); win[15]=red[draw()];
    if((p&15)==0&&!nostream){ acc(A_TC+(p>>4)*10,6); }
    uint32_t d=0; for(int k=0;k<8;k++) d=d*R+win[pos[k]]; if(d&7) continue; uint32_t B=d>>3; probes++;
    if(layout!=1) acc(A_BM+(B>>5)*4,0);         // 1 bit per code, separate prefix per 64/256
    else acc(A_BM+(uint64_t)(B>>5)*8,0);         // interleaved {bits,prefix}
    if(!(bm[B>>3]>>(B&7)&1)) continue;
    joins++;
    if(layout==0) acc(A_PF+(uint64_t)(B>>6)*4,1);
    else if(layout==2) acc(A_PF+(uint64_t)(B>>8)*4,1);
    long lo=0,hi=ng; while(lo<hi){long m=(lo+hi)/2; if(gB[m]<B)lo=m+1;else hi=m;} long g=lo;
    acc(A_SL+(uint64_t)g*slotB,2);
    if(!nostream){acc(A_TD+p,6); acc(A_TD+p+47,6);}
    for(uint32_t e=0;e<gcount[g];e++){ pairs++; uint32_t x=(uint32_t)keys[gstart[g]+e]; if(gcount[g]>1) acc(A_QL+(uint64_t)(gstart[g]+e)*4,3); acc(A_FD+x/2,4); acc(A_FD+x/2+23,4); }
  }
  const char*nm[8]={"bitmap","prefix","slot","qlist","fold","","stream",""};
  uint64_t tm=0; for(int k=0;k<7;k++){ if(hits[k]+miss[k]) printf("%-7s acc %10lu miss %10lu (%.1f%%)\n",nm[k],hits[k]+miss[k],miss[k],100.0*miss[k]/(hits[k]+miss[k])); if(k!=6)tm+=miss[k]; }
  double scale=3.0e8/NT*8; printf("probes/class %ld joins %ld pairs %ld | structure misses x8 classes scaled to 3e8 letters: %.3g ; stream misses scaled %.3g\n",probes,joins,pairs,tm*scale,miss[6]*scale);
}
