"""HBM-resident image store: the input side of the joint training step, MI355X first.

The reference produces every image tensor of a step on the host: positives in DataLoader workers (`cv2.imread -> ToPILImage ->
Resize((224, 224)) -> RandomHorizontalFlip -> ToTensor`, oe_h.py:700-712, 1463-1471) and -- where its time goes -- every image drawn
as a NEGATIVE synchronously in the training thread (`get_image`, oe_h.py:668-677, called from 980-983 and 1003-1007), followed by
`torch.stack` and `.to(device)`.  What a file determines is the resized uint8 image (150 528 bytes at 224 x 224); the flip, the
`/ 255` and the layout are arithmetic.  So:

  * the resized uint8 images live in HBM, `[capacity, H, W, 3]` (all of ETHEC: 7 GB of the MI355X's 288), filled on first touch;
  * one kernel per step (`lec_image_gather_u8`) builds the float batch from slot numbers: gather, mirror, `/ 255` -- bit-identical to
    ToTensor -- in the channels_last layout the backbone reads;
  * a miss is decoded off the training thread: by the DataLoader workers for the positives they were going to decode anyway
    (they hand over uint8 pixels, a quarter of the float tensor, and skip images a shared flag marks as resident), by this
    store's decode pool for negatives, which the trainer requests ONE STEP AHEAD (it knows step t+1's negatives: the sampler's
    stream is deterministic); all misses of a step travel in ONE pinned, asynchronous host-to-device copy.

File decoding itself stays PIL's (`decode_u8`); it is the one part of the reference's pipeline that is a third-party codec."""
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import _lib


class ImageRef:
    """What a store-backed dataset item carries in place of the float image tensor: the image's name, whether the train transform
    mirrored it, and -- when a DataLoader worker had to decode the file -- the resized uint8 pixels `[H, W, 3]` for the store."""
    __slots__ = ('name', 'flip', 'pixels')

    def __init__(self, name, flip=False, pixels=None):
        self.name, self.flip, self.pixels = name, bool(flip), pixels

    def __repr__(self):
        return 'ImageRef(%r, flip=%s%s)' % (self.name, self.flip, ', +pixels' if self.pixels is not None else '')


def decode_u8(loc, hw=224):
    """File -> resized uint8 `[hw, hw, 3]` in the reference's channel order.  The reference decodes with cv2.imread (B, G, R; 8 bit; no
    alpha) and gives the array to ToPILImage WITHOUT a channel swap, then Resize((hw, hw)) = PIL bilinear (oe_h.py:668-677, 700-712,
    1463-1471): tensor channel 0 is BLUE.  Resizing is per channel, so swapping before or after it is the same.  cv2 is not in this
    image: PIL decodes (libjpeg-turbo for JPEG; a different IDCT / chroma upsampling than cv2's build can move JPEG pixels by 1-2 / 255,
    PNG is lossless and identical)."""
    from PIL import Image
    with Image.open(loc) as im:
        pil = im.convert('RGB').resize((hw, hw), Image.BILINEAR)
    return np.ascontiguousarray(np.asarray(pil)[:, :, ::-1])


class StoreView:
    """The part of the store a (forked) DataLoader worker may look at: name -> id, and the shared 'resident in HBM' flags."""

    def __init__(self, index_of, flags, hw):
        self.index_of, self.flags, self.hw = index_of, flags, hw

    def is_resident(self, name):
        i = self.index_of.get(name)
        return i is not None and bool(self.flags[i])


class ImageStore:
    """uint8 image store in HBM + the host bookkeeping around it.

    locs: {name: path} of every image the store may be asked for (ids are positions in this dict).  capacity: slots in HBM (default: all
    of them; fewer: first-in first-out replacement, never of an image of the step being assembled).  decode_threads: size of the decode
    pool for misses nobody pre-decoded (PIL releases the GIL while it decodes and resizes)."""

    def __init__(self, locs, device, hw=224, capacity=None, decode_threads=4, decoder=None):
        self.device = torch.device(device)
        self.hw = int(hw)
        self.names = list(locs)
        self.locs = dict(locs)
        self.index_of = {n: i for i, n in enumerate(self.names)}
        n = len(self.names)
        self.capacity = max(1, min(n, int(capacity) if capacity else n))
        self.store = torch.empty((self.capacity, self.hw, self.hw, 3), dtype=torch.uint8, device=self.device)
        self.slot_of_id = np.full(n, -1, dtype=np.int64)
        self.id_of_slot = np.full(self.capacity, -1, dtype=np.int64)
        self.flags = torch.zeros(max(n, 1), dtype=torch.uint8).share_memory_()       # seen by forked DataLoader workers
        self._next = 0
        self._lock = threading.Lock()
        self._pending = {}                                       # id -> Future of a uint8 array (decode pool)
        self._offered = {}                                       # id -> uint8 array / tensor a worker decoded
        self._decoder = decoder or decode_u8
        self._pool = ThreadPoolExecutor(max_workers=max(1, int(decode_threads)), thread_name_prefix='lec-decode')
        self._staging = []                                       # ring of [pinned uint8 buffer, event that frees it]
        from .parallel import PinnedRing
        self._h2d = PinnedRing()                                 # slot numbers / flip bits of a step travel through pinned memory
        self.stats = {'hits': 0, 'decoded_here': 0, 'decoded_by_workers': 0, 'uploads': 0, 'upload_bytes': 0, 'evicted': 0}

    # ---- host side -----------------------------------------------------------------------------------------------------------------
    def view(self):
        return StoreView(self.index_of, self.flags, self.hw)

    def holds(self, name):
        return name in self.index_of

    def request(self, names):
        """Start decoding whatever of `names` is neither resident nor on its way (any thread; returns at once)."""
        with self._lock:
            for nm in names:
                i = self.index_of[nm]
                if self.slot_of_id[i] < 0 and i not in self._pending and i not in self._offered:
                    self._pending[i] = self._pool.submit(self._decoder, self.locs[nm], self.hw)

    def offer(self, name, pixels):
        """Pixels somebody else decoded (a DataLoader worker): kept until the next resolve() uploads them."""
        i = self.index_of[name]
        with self._lock:
            if self.slot_of_id[i] < 0 and i not in self._offered:
                self._offered[i] = pixels
                self.stats['decoded_by_workers'] += 1

    def _staging_buffer(self, m):
        for ent in self._staging:
            if ent[0].shape[0] >= m and ent[1].query():
                return ent
        if len(self._staging) >= 3:                              # all in flight or too small: wait for / replace the oldest
            ent = self._staging.pop(0)
            ent[1].synchronize()
        ent = [torch.empty((max(m, 64), self.hw, self.hw, 3), dtype=torch.uint8).pin_memory(), torch.cuda.Event()]
        ent[1].record()
        self._staging.append(ent)
        return ent

    def resolve(self, names):
        """Slots (numpy int32) of `names`, every one resident when the copies enqueued here on the CURRENT stream have run."""
        ids = np.fromiter((self.index_of[nm] for nm in names), dtype=np.int64, count=len(names))
        with self._lock:
            slots = self.slot_of_id[ids]
            miss = np.unique(ids[slots < 0])
            self.stats['hits'] += int(len(ids) - np.count_nonzero(slots < 0))
            if len(miss) == 0:
                return slots.astype(np.int32)
            for i in miss.tolist():                              # nobody asked ahead: decode now, in parallel
                if i not in self._pending and i not in self._offered:
                    self._pending[i] = self._pool.submit(self._decoder, self.locs[self.names[i]], self.hw)
            # (the entries stay in _pending / _offered until the slots below are assigned: a request() from the lookahead thread in between
            # must keep seeing these images as "on their way", or it decodes them a second time into a future nobody collects)
            futs = {i: self._pending[i] for i in miss.tolist() if i in self._pending}
            offered = {i: self._offered[i] for i in miss.tolist() if i in self._offered}
        arrays = {}
        for i, f in futs.items():
            arrays[i] = f.result()                               # (outside the lock: the lookahead thread keeps requesting)
            self.stats['decoded_here'] += 1
        arrays.update(offered)
        m = len(miss)
        if m > self.capacity:
            raise RuntimeError('image store of %d slots cannot hold the %d distinct new images of one step' % (self.capacity, m))
        ent = self._staging_buffer(m)
        stage_np = ent[0].numpy()
        for j, i in enumerate(miss.tolist()):
            a = arrays[i]
            a = a.numpy() if torch.is_tensor(a) else np.asarray(a)
            if a.shape != (self.hw, self.hw, 3) or a.dtype != np.uint8:
                raise ValueError('image %r decoded to %s %s, expected uint8 [%d, %d, 3]' % (self.names[i], a.dtype, a.shape, self.hw, self.hw))
            stage_np[j] = a
        with self._lock:
            keep = set(self.slot_of_id[ids[slots >= 0]].tolist())      # resident images of THIS step are never replaced
            new_slots = np.empty(m, dtype=np.int64)
            for j, i in enumerate(miss.tolist()):
                s = self._next % self.capacity
                while s in keep:
                    self._next += 1; s = self._next % self.capacity
                self._next += 1
                old = self.id_of_slot[s]
                if old >= 0:
                    self.slot_of_id[old] = -1; self.flags[old] = 0; self.stats['evicted'] += 1
                self.id_of_slot[s] = i; self.slot_of_id[i] = s; keep.add(s)
                new_slots[j] = s
                self._pending.pop(i, None); self._offered.pop(i, None)
        # one asynchronous copy per run of consecutive slots (a cold store: ONE copy for all misses of the step)
        a = 0
        while a < m:
            b = a + 1
            while b < m and new_slots[b] == new_slots[b - 1] + 1:
                b += 1
            s0 = int(new_slots[a])
            self.store[s0:s0 + (b - a)].copy_(ent[0][a:b], non_blocking=True)
            self.stats['uploads'] += 1
            a = b
        ent[1].record()
        self.stats['upload_bytes'] += m * self.hw * self.hw * 3
        self.flags[torch.from_numpy(miss)] = 1
        return self.slot_of_id[ids].astype(np.int32)

    # ---- device side ---------------------------------------------------------------------------------------------------------------
    def gather(self, slots, flips=None, c_out=3):
        """Float batch of the images in `slots` (numpy / list / device int32): fp32 `[n, c_out, H, W]` in channels_last memory,
        `uint8 / 255` (= ToTensor), mirrored along W where `flips` says so.  c_out = 4: zero 4th channel (the f32 stem's operand)."""
        ring = self._h2d
        ring.begin_step()
        if not torch.is_tensor(slots):
            slots = ring.upload(np.asarray(slots, dtype=np.int32), self.device)
        n = int(slots.shape[0])
        out = torch.empty((n, c_out, self.hw, self.hw), dtype=torch.float32, device=self.device, memory_format=torch.channels_last)
        if n == 0:
            return out
        fl = None
        if flips is not None:
            if not torch.is_tensor(flips):
                flips = np.asarray(flips, dtype=np.uint8)
                fl = ring.upload(flips, self.device) if flips.any() else None
            else:
                fl = flips.to(torch.uint8)
        ring.end_step()
        _lib.check(_lib.lib.lec_image_gather_u8(_lib.dptr(self.store), self.capacity, _lib.dptr(slots), _lib.dptr(fl), n, self.hw, self.hw,
                                                c_out, _lib.dptr(out), _lib.stream_ptr()))
        return out

    def batch(self, names, flips=None, c_out=3):
        """resolve + gather: the float batch for `names`, decoding / uploading what is not resident yet."""
        return self.gather(self.resolve(names), flips, c_out)

    def close(self):
        self._pool.shutdown(wait=False, cancel_futures=True)
