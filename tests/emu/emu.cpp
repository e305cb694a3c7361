// emu.cpp -- scheduler of the fiber-based SIMT emulator (see emu.h).  TEST INFRASTRUCTURE ONLY.
#include "emu.h"
#include <assert.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

thread_local EmuBlockCtx* emu_blk = nullptr;
thread_local EmuLaneState* emu_cur = nullptr;

// Minimal x86-64 SysV context switch: callee-saved registers + stack pointer.
extern "C" void emu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size emu_switch,.-emu_switch
)");

static thread_local EmuKernelBody g_body;
static thread_local void* g_args;
static thread_local uint8_t* g_smem;

static void lane_entry() {
    g_body(g_args, g_smem);
    EmuLaneState* me = emu_cur;
    me->done = true;
    me->op = EMU_EXIT;
    emu_switch(&me->sp, emu_blk->sched_sp);
    abort();  // never resumed
}

sky_u64 emu_collective(int op, sky_u64 a0, sky_u64 a1) {
    EmuLaneState* me = emu_cur;
    me->op = op; me->arg0 = a0; me->arg1 = a1;
    emu_switch(&me->sp, emu_blk->sched_sp);
    return me->result;
}

static const size_t STACK_BYTES = 256 * 1024;

static void resume(EmuLaneState* l) {
    emu_cur = l;
    l->op = EMU_NONE;
    emu_switch(&emu_blk->sched_sp, l->sp);
    emu_cur = nullptr;
}

void emu_launch(int grid, int block, size_t lds_bytes, EmuKernelBody body, void* args) {
    assert(block % 64 == 0 && block <= 1024);
    const int nw = block / 64;
    std::vector<EmuLaneState> lanes(block);
    std::vector<uint8_t> smem(lds_bytes + 16);
    for (int t = 0; t < block; t++) lanes[t].stack = aligned_alloc(64, STACK_BYTES);
    EmuBlockCtx ctx;
    ctx.bdim = block; ctx.gdim = grid; ctx.lanes = lanes.data();
    g_body = body; g_args = args; g_smem = smem.data();
    for (int b = 0; b < grid; b++) {
        ctx.bid = b;
        emu_blk = &ctx;
        memset(smem.data(), 0xA5, smem.size());   // LDS is uninitialised on hardware: poison it
        for (int t = 0; t < block; t++) {
            EmuLaneState& l = lanes[t];
            l.tid = t; l.done = false; l.op = EMU_NONE; l.result = 0;
            // initial frame: 6 callee-saved slots + return address = lane_entry; entry rsp % 16 == 8
            uintptr_t top = ((uintptr_t)l.stack + STACK_BYTES) & ~(uintptr_t)15;
            void** sp = (void**)(top - 8);      // fake return address slot of lane_entry's "caller"
            *--sp = (void*)lane_entry;          // popped by ret
            for (int k = 0; k < 6; k++) *--sp = nullptr;
            l.sp = sp;
        }
        std::vector<char> at_barrier(nw, 0), finished(nw, 0);
        for (;;) {
            bool all_done = true;
            for (int w = 0; w < nw; w++) {
                if (finished[w] || at_barrier[w]) { if (!finished[w]) all_done = false; continue; }
                all_done = false;
                EmuLaneState* wl = &lanes[w * 64];
                for (;;) {
                    // run every live lane to its next collective (or exit)
                    for (int i = 0; i < 64; i++) if (!wl[i].done) resume(&wl[i]);
                    int op = EMU_NONE; int live = 0;
                    for (int i = 0; i < 64; i++) if (!wl[i].done) {
                        live++;
                        if (op == EMU_NONE) op = wl[i].op;
                        if (wl[i].op != op) { fprintf(stderr, "emu: divergent collective in block %d wave %d: lane %d op %d vs %d\n", b, w, i, wl[i].op, op); abort(); }
                    }
                    if (!live) { finished[w] = 1; break; }
                    if (op == EMU_BARRIER) { at_barrier[w] = 1; break; }
                    if (op == EMU_YIELD) break;      // resumed on the scheduler's next pass over the wavefronts
                    switch (op) {
                    case EMU_BALLOT: {
                        sky_u64 m = 0;
                        for (int i = 0; i < 64; i++) if (!wl[i].done && wl[i].arg0) m |= 1ull << i;
                        for (int i = 0; i < 64; i++) wl[i].result = m;
                    } break;
                    case EMU_READLANE: {
                        int src = (int)(long long)wl[0].arg1; bool first = false;
                        for (int i = 0; i < 64; i++) if (!wl[i].done) {
                            int s = (int)(long long)wl[i].arg1;
                            if (!first) { src = s; first = true; }
                            if (s != src) { fprintf(stderr, "emu: readlane with non-uniform lane index\n"); abort(); }
                        }
                        if (src < 0) { for (int i = 0; i < 64; i++) if (!wl[i].done) { src = i; break; } }
                        assert(src >= 0 && src < 64);
                        sky_u64 v = wl[src].done ? 0 : wl[src].arg0;
                        for (int i = 0; i < 64; i++) wl[i].result = v;
                    } break;
                    case EMU_SHFL:
                        for (int i = 0; i < 64; i++) if (!wl[i].done) {
                            int s = (int)wl[i].arg1 & 63;
                            wl[i].result = wl[s].done ? 0 : wl[s].arg0;
                        }
                        break;
                    case EMU_PUSH: {     // ds_permute_b32: a lane nobody sends to receives 0, the highest sender wins
                        sky_u64 got[64] = {0};
                        for (int i = 0; i < 64; i++) if (!wl[i].done) got[(int)wl[i].arg1 & 63] = wl[i].arg0;
                        for (int i = 0; i < 64; i++) wl[i].result = got[i];
                    } break;
                    case EMU_SCAN: {
                        uint32_t acc = 0;
                        for (int i = 0; i < 64; i++) { if (!wl[i].done) acc += (uint32_t)wl[i].arg0; wl[i].result = acc; }
                    } break;
                    case EMU_SCANMAX: {
                        uint32_t acc = 0;
                        for (int i = 0; i < 64; i++) { if (!wl[i].done && (uint32_t)wl[i].arg0 > acc) acc = (uint32_t)wl[i].arg0; wl[i].result = acc; }
                    } break;
                    case EMU_WAVESYNC: break;
                    default: fprintf(stderr, "emu: bad op %d\n", op); abort();
                    }
                }
            }
            if (all_done) break;
            bool release = true;
            for (int w = 0; w < nw; w++) if (!finished[w] && !at_barrier[w]) release = false;
            if (release) for (int w = 0; w < nw; w++) at_barrier[w] = 0;
        }
    }
    emu_blk = nullptr;
    for (int t = 0; t < block; t++) free(lanes[t].stack);
}
