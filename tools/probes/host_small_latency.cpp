// Host-core latency of the m x m work of one restart (m = 40, 18 shifts): TridiagEigen (values + last row / full) and the shifted QR
// sweeps, the same routines the library runs (include/Spectra/internal/SmallDense.h).  g++ -O2 -std=c++17 -Iinclude tools/probes/host_small_latency.cpp
#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <random>
#include "Spectra/internal/SmallDense.h"
using namespace mispec::small;
int main(){
  const int m=40, nshift=18;
  std::mt19937_64 g(1); std::uniform_real_distribution<double> U(-0.5,0.5);
  std::vector<double> d0(m), e0(m);
  for(int i=0;i<m;i++){ d0[i]=4*U(g); e0[i]=1.0+U(g);} e0[m-1]=0;
  double best_e=1e9,best_q=1e9,best_ef=1e9; double sink=0;
  for(int rep=0;rep<200;rep++){
    std::vector<double> d=d0,e=e0,Q(m*m,0.0); for(int i=0;i<m;i++)Q[i*m+i]=1;
    auto t0=std::chrono::steady_clock::now();
    tridiag_eigen(m,d.data(),e.data(),Q.data(),m,Lanes{m-1,1});
    auto t1=std::chrono::steady_clock::now();
    best_e=std::min(best_e,std::chrono::duration<double,std::micro>(t1-t0).count());
    std::vector<double> ev=d; std::sort(ev.begin(),ev.end(),[](double a,double b){return std::fabs(a)>std::fabs(b);});
    std::vector<double> d2=d0,e2=e0,Q2(m*m,0.0),work(4*m); for(int i=0;i<m;i++)Q2[i*m+i]=1;
    t0=std::chrono::steady_clock::now();
    for(int s=0;s<nshift;s++) tridiag_shifted_qr(m,d2.data(),e2.data(),ev[m-nshift+s],Q2.data(),m,m,work.data(),Lanes{0,1});
    t1=std::chrono::steady_clock::now();
    best_q=std::min(best_q,std::chrono::duration<double,std::micro>(t1-t0).count());
    sink+=Q2[5]+d2[3];
    d=d0;e=e0; std::fill(Q.begin(),Q.end(),0.0); for(int i=0;i<m;i++)Q[i*m+i]=1;
    t0=std::chrono::steady_clock::now();
    tridiag_eigen(m,d.data(),e.data(),Q.data(),m,Lanes{0,1});
    t1=std::chrono::steady_clock::now();
    best_ef=std::min(best_ef,std::chrono::duration<double,std::micro>(t1-t0).count());
    sink+=Q[7];
  }
  printf("eigen(last row) %.1f us, eigen(full) %.1f us, %d shifted QR sweeps %.1f us (sink %g)\n",best_e,best_ef,nshift,best_q,sink);
}
