#include "pchain.h"
#include <vector>
#include <cstdio>
namespace blvm { void set_error(const char*, ...) {} }
using blvm::pchain::TileIter;
int main() {
  int bad = 0;
  for (int xcd = 0; xcd < 2; ++xcd)
    for (int rt : {1, 3, 4, 8})
      for (int ct : {1, 12, 16, 32, 96})
        for (int wg0 : {0, 128, 64})
          for (int nwg : {8, 16, 64, 128})
            for (int grid : {256}) {
              std::vector<int> cnt(rt * ct, 0);
              for (int w = 0; w < grid; ++w)
                for (TileIter it(w, wg0, nwg, rt, ct, xcd); it.valid(); it.next()) {
                  int r = it.r0() / 16, c = it.c();
                  if (w < wg0 || w >= wg0 + nwg || r < 0 || r >= rt || c < 0 || c >= ct) { ++bad; continue; }
                  cnt[c * rt + r]++;
                }
              for (int v : cnt) bad += v != 1;
            }
  printf("tile iterator: %d errors\n", bad);
  return bad != 0;
}
