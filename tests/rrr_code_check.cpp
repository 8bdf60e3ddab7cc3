// CPU check of the block code of the rrr-63 index variant (vlg_matching_amd/csrc/rrr_code.hpp), built and run by
// tests/test_rrr_code.py: every class is numbered 0 .. C(63,k)-1 without gaps (checked exhaustively for the classes that are
// small enough), offsets fit the widths rrr_vector<63> gives them, and decode(encode(x)) returns the ones before every bit
// position and the bit itself.
#include "rrr_code.hpp"
#include <cstdio>
#include <random>
#include <vector>
#include <algorithm>
using namespace vlg;
int main(){
    static RrrTables t; build_rrr_tables(t);
    printf("sizeof %zu\n", sizeof t);
    std::mt19937_64 rng(7);
    // exhaustive classes 0..3 and 60..63
    for (int k : {0,1,2,3,60,61,62,63}) {
        std::vector<uint64_t> offs;
        std::vector<int> pos(k);
        // enumerate combos of zeros/ones
        int m = k <= 3 ? k : 63 - k; bool inv = k > 3;
        std::vector<int> c(m); for (int i=0;i<m;++i) c[i]=i;
        for(;;){
            uint64_t x=0; for(int i=0;i<m;++i) x|=1ull<<c[i];
            if(inv) x = ~x & ((1ull<<63)-1);
            uint32_t kk; uint64_t o = rrr_enc63(t,x,kk);
            if((int)kk!=k){printf("class mismatch\n");return 1;}
            offs.push_back(o);
            int i=m-1; while(i>=0 && c[i]==63-m+i) --i; if(i<0) break; ++c[i]; for(int j=i+1;j<m;++j) c[j]=c[j-1]+1;
            if(m==0) break;
        }
        std::sort(offs.begin(),offs.end());
        for(size_t i=0;i<offs.size();++i) if(offs[i]!=i){printf("not a permutation k=%d at %zu\n",k,i);return 1;}
        printf("class %d: %zu blocks numbered 0..%zu\n",k,offs.size(),offs.size()-1);
    }
    // random: all densities
    uint64_t checks=0;
    for(int it=0; it<100000; ++it){
        uint64_t x=rng() & ((1ull<<63)-1);
        int mode=it%5; if(mode==1) x&=rng(); if(mode==2) x|=rng()&((1ull<<63)-1); if(mode==3) x&=rng()&rng()&rng(); if(mode==4) x = (x|rng()|rng()) & ((1ull<<63)-1);
        uint32_t k; uint64_t o=rrr_enc63(t,x,k);
        if(k!=(uint32_t)__builtin_popcountll(x)){printf("k\n");return 1;}
        uint32_t sp=t.space[k]; if(sp<64 && (o>>sp)){printf("offset too wide k=%u\n",k);return 1;}
        for(uint32_t off=0; off<63; ++off){
            uint32_t bit; uint32_t ones=rrr_dec63(t,k,o,off,bit);
            uint32_t want=__builtin_popcountll(x & ((1ull<<off)-1)), wb=(x>>off)&1;
            if(ones!=want||bit!=wb){printf("mismatch x=%llx k=%u off=%u got %u/%u want %u/%u\n",(unsigned long long)x,k,off,ones,bit,want,wb);return 1;}
            ++checks;
        }
    }
    printf("ok %llu checks\n",(unsigned long long)checks);
}
