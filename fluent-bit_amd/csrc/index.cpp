// index.cpp -- host side of the device record indexer (kernels: index_kernels.inc)
#include "host_int.hpp"

using namespace flbgpu;

struct flbgpu_indexer {
    hipStream_t stream = nullptr;
    DevBuf masks, tile_cnt, tile_off, scan_tmp, cand_pos, rec_len, succ, exitp, entry, flags, off, extras, state, row_off;
    PinnedBuf hstate;
    uint64_t last_candidates = 0, last_extras = 0, last_rounds = 0;
    ~flbgpu_indexer() {
        DevBuf *all[] = {&masks, &tile_cnt, &tile_off, &scan_tmp, &cand_pos, &rec_len, &succ, &exitp, &entry, &flags, &off, &extras, &state, &row_off};
        for (auto *b : all) b->release();
        hstate.release();
        if (stream) (void) hipStreamDestroy(stream);
    }
};

static const uint32_t EXTRAS_CAP = 1u << 20;
static const int MAX_ROUNDS = 1 << 16;

extern "C" flbgpu_indexer *flbgpu_indexer_create(void) {
    flbgpu_indexer *ix = new flbgpu_indexer();
    if (hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking) != hipSuccess) {
        set_err("hipStreamCreate failed (no HIP device?)");
        delete ix;
        return nullptr;
    }
    return ix;
}

extern "C" void flbgpu_indexer_destroy(flbgpu_indexer *ix) { delete ix; }

static bool index_dev_impl(flbgpu_indexer *ix, const uint8_t *data, size_t bytes, flbgpu_dev_chunk *out, size_t *consumed, int64_t *n_out) {
    hipStream_t st = ix->stream;
    const size_t tiles = idx_tiles(bytes);
    if (!ix->hstate.ensure(sizeof(IdxState) + 2 * sizeof(uint64_t))) return false;
    IdxState &hs = *ix->hstate.as<IdxState>();
    uint64_t &h_count = *(uint64_t *) (ix->hstate.as<uint8_t>() + sizeof(IdxState));
    if (!ix->masks.ensure((bytes / 64 + 2) * sizeof(uint64_t)) || !ix->tile_cnt.ensure(tiles * sizeof(uint32_t) + 4) || !ix->tile_off.ensure((tiles + 1) * sizeof(uint64_t)) ||
        !ix->scan_tmp.ensure(scan_tmp_elems(tiles) * sizeof(uint64_t)) || !ix->state.ensure(sizeof(IdxState)) ||
        !ix->extras.ensure((size_t) EXTRAS_CAP * sizeof(uint64_t)))
        return false;
    // 1. candidates
    launch_idx_count(data, bytes, ix->masks.as<uint64_t>(), ix->tile_cnt.as<uint32_t>(), st);
    launch_scan(ix->tile_cnt.as<uint32_t>(), tiles, ix->scan_tmp.as<uint64_t>(), ix->tile_off.as<uint64_t>(), st);
    HIPOK(hipMemcpyAsync(&h_count, ix->tile_off.as<uint64_t>() + tiles, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    HIPOK(hipStreamSynchronize(st));
    const uint64_t nc = h_count;
    if (nc >= 0xffffff00ull) { set_err("chunk has too many record candidates for one call (%llu)", (unsigned long long) nc); return false; }
    const size_t nblk = idx_blocks(nc);
    if (!ix->cand_pos.ensure((nc + 1) * sizeof(uint64_t)) || !ix->rec_len.ensure((nc + 1) * sizeof(uint32_t)) ||
        !ix->succ.ensure((nc + 1) * sizeof(uint32_t)) || !ix->exitp.ensure((nc + 1) * sizeof(uint32_t)) ||
        !ix->flags.ensure((nc + 1) * sizeof(uint32_t)) || !ix->off.ensure((nc + 2) * sizeof(uint64_t)) ||
        !ix->entry.ensure((nblk + 1) * sizeof(uint32_t)) || !ix->scan_tmp.ensure(scan_tmp_elems(nc > tiles ? nc : tiles) * sizeof(uint64_t)))
        return false;
    // 2. one speculative skip per candidate, 3. the part of the succ graph inside each block
    launch_idx_fill(ix->masks.as<uint64_t>(), bytes, ix->tile_off.as<uint64_t>(), ix->cand_pos.as<uint64_t>(), st);
    launch_idx_walk(data, bytes, ix->cand_pos.as<uint64_t>(), nc, ix->rec_len.as<uint32_t>(), ix->succ.as<uint32_t>(), st);
    launch_idx_exit(ix->succ.as<uint32_t>(), nc, ix->exitp.as<uint32_t>(), st);
    HIPOK(hipMemsetAsync(ix->flags.p, 0, (nc + 1) * sizeof(uint32_t), st));
    memset(&hs, 0, sizeof(hs));
    hs.kind = 0xffffffffu;
    hs.status = 3;                                         // k_idx_one decides how the walk starts at byte 0
    HIPOK(hipMemcpyAsync(ix->state.p, &hs, sizeof(hs), hipMemcpyHostToDevice, st));
    // Every round is: k_idx_one puts the walk on a candidate, the chain kernels follow it to where it
    // ends.  A clean chunk needs one; rounds queued after the walk has finished do nothing, so the
    // later ones are enqueued several per synchronisation.
    int rounds = 0;
    do {
        const int batch = rounds == 0 ? 1 : 16;
        for (int r = 1; r < batch; r++) {
            launch_idx_one(data, bytes, ix->cand_pos.as<uint64_t>(), nc, ix->extras.as<uint64_t>(), EXTRAS_CAP, ix->state.as<IdxState>(), st);
            launch_idx_chain_mark(ix->succ.as<uint32_t>(), ix->rec_len.as<uint32_t>(), ix->cand_pos.as<uint64_t>(), nc, ix->exitp.as<uint32_t>(),
                                  ix->entry.as<uint32_t>(), ix->flags.as<uint32_t>(), bytes, ix->state.as<IdxState>(), st);
            rounds++;
        }
        launch_idx_one(data, bytes, ix->cand_pos.as<uint64_t>(), nc, ix->extras.as<uint64_t>(), EXTRAS_CAP, ix->state.as<IdxState>(), st);
        launch_idx_chain_mark(ix->succ.as<uint32_t>(), ix->rec_len.as<uint32_t>(), ix->cand_pos.as<uint64_t>(), nc, ix->exitp.as<uint32_t>(),
                              ix->entry.as<uint32_t>(), ix->flags.as<uint32_t>(), bytes, ix->state.as<IdxState>(), st);
        HIPOK(hipMemcpyAsync(&hs, ix->state.p, sizeof(hs), hipMemcpyDeviceToHost, st));
        HIPOK(hipStreamSynchronize(st));
        rounds++;
    } while (hs.status == 3 && rounds < MAX_ROUNDS);
    ix->last_rounds = (uint64_t) rounds;
    if (hs.status != 1) {
        set_err(hs.status == 2 ? "record indexer: more than %u records do not start with a 2-element array"
                               : "record indexer: chain did not close (status %u)", hs.status == 2 ? EXTRAS_CAP : hs.status);
        return false;
    }
    // 4. compact
    launch_scan(ix->flags.as<uint32_t>(), nc, ix->scan_tmp.as<uint64_t>(), ix->off.as<uint64_t>(), st);
    HIPOK(hipMemcpyAsync(&h_count, ix->off.as<uint64_t>() + nc, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    HIPOK(hipStreamSynchronize(st));
    const uint64_t n = h_count + hs.n_extras;
    if (!ix->row_off.ensure((n + 1) * sizeof(uint64_t))) return false;
    launch_idx_emit(ix->cand_pos.as<uint64_t>(), nc, ix->flags.as<uint32_t>(), ix->off.as<uint64_t>(), ix->extras.as<uint64_t>(), hs.n_extras,
                    hs.consumed, ix->row_off.as<uint64_t>(), st);
    HIPOK(hipStreamSynchronize(st));
    ix->last_candidates = nc;
    ix->last_extras = hs.n_extras;
    out->data = data; out->row_off = ix->row_off.as<uint64_t>(); out->n = n; out->bytes = hs.consumed;
    *consumed = (size_t) hs.consumed;
    *n_out = (int64_t) n;
    return true;
}

extern "C" int64_t flbgpu_index_dev(flbgpu_indexer *ix, const void *dev_data, size_t bytes, flbgpu_dev_chunk *out, size_t *consumed) {
    flbgpu_dev_chunk tmp;
    size_t c = 0;
    int64_t n = 0;
    if (!out) out = &tmp;
    if (bytes == 0) {
        memset(out, 0, sizeof(*out));
        out->data = dev_data;
        if (consumed) *consumed = 0;
        return 0;
    }
    if (!index_dev_impl(ix, (const uint8_t *) dev_data, bytes, out, &c, &n)) return -1;
    if (consumed) *consumed = c;
    return n;
}

extern "C" void flbgpu_indexer_stats(const flbgpu_indexer *ix, uint64_t *candidates, uint64_t *off_chain_rows, uint64_t *rounds) {
    if (candidates) *candidates = ix->last_candidates;
    if (off_chain_rows) *off_chain_rows = ix->last_extras;
    if (rounds) *rounds = ix->last_rounds;
}
