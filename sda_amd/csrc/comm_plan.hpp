// Cross-GPU modular reduction of per-clerk partial sums (no reference counterpart - the reference's
// parties meet over HTTP).  The choreography, independent of what moves the bytes:
//
//   direct reduce-scatter : rank r sends slice g of its partial vector to rank g, for every g (on the xGMI mesh every
//                           pair of GPUs has its own link, so all 7 links of a GPU are busy at once; a ring would be
//                           bound by one link)
//   local modular sum     : rank g adds the G slices it now holds, exactly (128-bit), and reduces mod q
//   direct all-gather     : rank g sends its reduced slice to every rank, straight into place
//
// A plain sum collective on u64 is NOT usable: 8 residues of a 62-bit modulus exceed 2^64.  Slices are exact-sized
// (the last one may be shorter or empty): nothing is padded and nothing is staged.
//
// The transport and the reducer are interfaces: the product binds them to RCCL point-to-point calls and to the HIP
// kernel (sda_comm.cpp); tests/cpp/comm_plan_test.cpp binds them to socket pairs between forked processes and a
// checker, which is how the choreography is covered on a box without GPUs.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace sda {

struct SlicePlan {
    int world;
    size_t len, seg;                                           // seg = ceil(len / world)
    SlicePlan(int w, size_t n) : world(w), len(n), seg(w > 0 ? (n + (size_t)w - 1) / (size_t)w : 0) {}
    size_t offset(int g) const { const size_t o = (size_t)g * seg; return o < len ? o : len; }
    size_t count(int g) const { return offset(g + 1) - offset(g); }          // offset() clamps at len
};

struct Transport {                      // grouped point-to-point operations on int64 elements
    virtual int group_start() = 0;
    virtual int send(const int64_t* buf, size_t count, int peer) = 0;
    virtual int recv(int64_t* buf, size_t count, int peer) = 0;
    virtual int group_end() = 0;        // every operation posted since group_start is complete (or enqueued in order)
    virtual ~Transport() {}
};

struct Reducer {                        // out[i] = sum over g < parts of in[g * stride + i]  mod q, i < len
    virtual int modsum(const int64_t* in, size_t parts, size_t stride, size_t len, int64_t* out) = 0;
    virtual ~Reducer() {}
};

// partial[len] (this rank's canonical residues) -> out[len] = sum over ranks mod q, on every rank.
// recv_scratch: world * seg elements; mine_scratch: seg elements.  Returns the first non-zero status.
inline int modular_allreduce_plan(Transport& tr, Reducer& red, int rank, int world, const int64_t* partial, size_t len,
                                  int64_t* recv_scratch, int64_t* mine_scratch, int64_t* out) {
    const SlicePlan pl(world, len);
    const size_t mine = pl.count(rank);
    // A send / recv that fails between group_start and group_end must not leave the group open (an open ncclGroupStart on
    // the thread makes the next RCCL call misbehave): the group is always closed, best effort, and the FIRST error wins.
    auto grouped = [&](auto&& post) -> int {
        int st = tr.group_start();
        if (st) return st;
        st = post();
        const int end = tr.group_end();
        return st ? st : end;
    };
    // reduce-scatter: slice g of every rank meets on rank g
    int st = grouped([&]() -> int {
        for (int g = 0; g < world; ++g) {
            if (pl.count(g)) if (int e = tr.send(partial + pl.offset(g), pl.count(g), g)) return e;
            if (mine) if (int e = tr.recv(recv_scratch + (size_t)g * pl.seg, mine, g)) return e;
        }
        return 0;
    });
    if (st) return st;
    if (mine && (st = red.modsum(recv_scratch, (size_t)world, pl.seg, mine, mine_scratch))) return st;
    // all-gather: reduced slice g goes from rank g to everybody, straight into place
    return grouped([&]() -> int {
        for (int g = 0; g < world; ++g) {
            if (mine) if (int e = tr.send(mine_scratch, mine, g)) return e;
            if (pl.count(g)) if (int e = tr.recv(out + pl.offset(g), pl.count(g), g)) return e;
        }
        return 0;
    });
}

}  // namespace sda
