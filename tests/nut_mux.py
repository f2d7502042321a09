"""Test infrastructure: the byte stream `ffmpeg ... -vcodec rawvideo -f nut md5:` produces for ONE raw video stream, restated from the reference's
NUT muxer so that the reference's own `filter-pixfmts-*` / `filter-pixdesc-*` known answers (tests/ref/fate/*: MD5s of whole NUT files, the only
goldens it holds for scaled output per pixel format and for sws_scale_frame()'s property-driven mode) can be reproduced without libavformat.

Restates, for the one-video-stream case FATE's video_filter() produces (tests/fate-run.sh:621-629):
  libavformat/nutenc.c  choose_timebase :50-61, build_frame_code :167-309, put_v / put_tt / put_str / put_s :314-359, put_packet :361-381,
                        write_mainheader :383-448, write_streamheader :450-503, write_globalinfo :514-541, write_streaminfo :543-583,
                        write_index :611-661, write_headers :663-714, nut_write_header :716-803, get_needed_flags :805-833,
                        nut_write_packet :959-1175, nut_write_trailer :1177-1198
  libavformat/nut.h     start codes :29-33, ID_STRING :35, MAX_DISTANCE :37, flags :43-56
  libavutil/crc.c       AV_CRC_32_IEEE (polynomial 0x04C11DB7, MSB first, no final xor); the muxer stores it most significant byte first
  libavcodec/rawenc.c   raw_encode :49-83: av_image_copy_to_buffer(..., align 1): the planes' visible rows back to back
Nothing of the product or of the oracle is used here; tests/test_*_fate_nut.py feed it frames and compare MD5s."""
import hashlib

MAIN_STARTCODE = 0x7A561F5F04AD + ((ord('N') << 8) + ord('M') << 48)
STREAM_STARTCODE = 0x11405BF2F9DB + ((ord('N') << 8) + ord('S') << 48)
SYNCPOINT_STARTCODE = 0xE4ADEECA4569 + ((ord('N') << 8) + ord('K') << 48)
INDEX_STARTCODE = 0xDD672F23E64E + ((ord('N') << 8) + ord('X') << 48)
INFO_STARTCODE = 0xAB68B596BA78 + ((ord('N') << 8) + ord('I') << 48)
ID_STRING = b"nut/multimedia container\0"
MAX_DISTANCE = 1024 * 32 - 1
FLAG_KEY, FLAG_EOR, FLAG_CODED_PTS, FLAG_STREAM_ID, FLAG_SIZE_MSB, FLAG_CHECKSUM = 1, 2, 8, 16, 32, 64
FLAG_SM_DATA, FLAG_HEADER_IDX, FLAG_MATCH_TIME, FLAG_CODED, FLAG_INVALID = 256, 1024, 2048, 4096, 8192
NOPTS = -(1 << 63)

_CRC_TAB = []
for _i in range(256):
    _c = _i << 24
    for _ in range(8):
        _c = ((_c << 1) ^ 0x04C11DB7) & 0xFFFFFFFF if _c & 0x80000000 else (_c << 1) & 0xFFFFFFFF
    _CRC_TAB.append(_c)


def crc32_ieee_msb(data, crc=0):
    for b in data:
        crc = ((crc << 8) & 0xFFFFFFFF) ^ _CRC_TAB[(crc >> 24) ^ b]
    return crc


def v_length(val):
    i = 1
    val >>= 7
    while val:
        i += 1
        val >>= 7
    return i


def put_v(val):
    val &= (1 << 64) - 1                      # uint64_t argument
    n = v_length(val)
    return bytes([0x80 | ((val >> (7 * i)) & 0x7F) for i in range(n - 1, 0, -1)] + [val & 0x7F])


def put_s(val):
    return put_v(2 * abs(val) - (1 if val > 0 else 0))


def put_str(s):
    b = s.encode() if isinstance(s, str) else s
    return put_v(len(b)) + b


def put_packet(payload, startcode):
    forw_ptr = len(payload) + 4
    head = startcode.to_bytes(8, "big") + put_v(forw_ptr)
    out = head
    if forw_ptr > 4096:
        out += crc32_ieee_msb(head).to_bytes(4, "big")
    return out + payload + crc32_ieee_msb(payload).to_bytes(4, "big")


def choose_timebase(num, den, min_precision):
    j = 2
    while j < 14:
        while den // num < min_precision and num % j == 0:
            num //= j
        j += 1 + (1 if j > 2 else 0)
    while den // num < min_precision and den < (1 << 24):
        den <<= 1
    return num, den


class NutVideoMuxer:
    """one raw video stream; packets in presentation order"""

    def __init__(self, width, height, codec_tag, time_base=(1, 25), frame_rate=(25, 1), stream_metadata=(("encoder", "Lavc rawvideo"),), sar=(0, 1)):
        self.w, self.h, self.tag = width, height, codec_tag
        self.frame_rate, self.meta, self.sar = frame_rate, list(stream_metadata), sar
        self.tb = choose_timebase(time_base[0], time_base[1], 48000)
        self.in_tb = time_base
        self.msb_pts_shift = 7 if 1000 * self.tb[0] >= self.tb[1] else 14
        self.max_pts_distance = max(self.tb[1], self.tb[0]) // self.tb[0]
        self.header_len = [0, 3, 4, 2, 2, 2, 2]     # build_elision_headers: index 0 = none
        self.header_count = 7
        self._build_frame_code()
        self.out = bytearray()
        self.last_syncpoint_pos = -(1 << 31)
        self.last_flags = 0
        self.last_pts = NOPTS
        self.sp = []                                 # syncpoint positions
        self.keyframe_pts = []
        self.max_pts = None
        self.index_entries = []                      # (pos, pts) of key frames

    def _build_frame_code(self):
        fc = [dict(flags=0, pts_delta=0, size_mul=0, size_lsb=0, stream_id=0, header_idx=0) for _ in range(256)]
        start, end = 1, 254
        fc[start].update(flags=FLAG_CODED, size_mul=1, pts_delta=1)
        start += 1
        # one video stream: the whole range is its
        start2, end2 = start, start + (end - start)
        # f = (1 / frame_rate) / time_base
        fn, fd = self.frame_rate[1] * self.tb[1], self.frame_rate[0] * self.tb[0]
        from math import gcd
        g = gcd(fn, fd)
        fn, fd = fn // g, fd // g
        frame_size = fn if (fd == 1 and fn > 0) else 1
        for key_frame in (0, 1):
            fc[start2].update(flags=FLAG_KEY * key_frame | FLAG_SIZE_MSB | FLAG_CODED_PTS, stream_id=0, size_mul=1)
            start2 += 1
        key_frame = 0
        fc[start2].update(flags=FLAG_KEY | FLAG_SIZE_MSB, stream_id=0, size_mul=1, pts_delta=frame_size)
        start2 += 1
        pred_table = [1]
        for pred in range(len(pred_table)):
            start3 = start2 + (end2 - start2) * pred // len(pred_table)
            end3 = start2 + (end2 - start2) * (pred + 1) // len(pred_table)
            pred_table[pred] *= frame_size
            for index in range(start3, end3):
                fc[index].update(flags=FLAG_KEY * key_frame | FLAG_SIZE_MSB, stream_id=0, size_mul=end3 - start3, size_lsb=index - start3, pts_delta=pred_table[pred])
        N = ord('N')
        fc[N + 1:256] = [dict(e) for e in fc[N:255]]          # memmove(&frame_code['N' + 1], &frame_code['N'], sizeof(FrameCode) * (255 - 'N'))
        for i in (0, 255, N):
            fc[i]["flags"] = FLAG_INVALID
        self.fc = fc

    # ---- headers ----
    def _mainheader(self):
        b = bytearray()
        b += put_v(3)                                # NUT_STABLE_VERSION
        b += put_v(1)                                # nb_streams
        b += put_v(MAX_DISTANCE)
        b += put_v(1)                                # time_base_count
        b += put_v(self.tb[0]) + put_v(self.tb[1])
        tmp_pts, tmp_mul, tmp_stream, tmp_head_idx = 0, 1, 0, 0
        tmp_match = 1 - (1 << 62)
        fc = self.fc
        i = 0
        N = ord('N')
        while i < 256:
            tmp_fields, tmp_size = 0, 0
            if tmp_pts != fc[i]["pts_delta"]: tmp_fields = 1
            if tmp_mul != fc[i]["size_mul"]: tmp_fields = 2
            if tmp_stream != fc[i]["stream_id"]: tmp_fields = 3
            if tmp_size != fc[i]["size_lsb"]: tmp_fields = 4
            if tmp_head_idx != fc[i]["header_idx"]: tmp_fields = 8
            tmp_pts, tmp_flags, tmp_stream = fc[i]["pts_delta"], fc[i]["flags"], fc[i]["stream_id"]
            tmp_mul, tmp_size, tmp_head_idx = fc[i]["size_mul"], fc[i]["size_lsb"], fc[i]["header_idx"]
            j = 0
            while i < 256:
                if i == N:
                    i += 1
                    continue
                e = fc[i]
                if (e["pts_delta"] != tmp_pts or e["flags"] != tmp_flags or e["stream_id"] != tmp_stream or e["size_mul"] != tmp_mul or
                        e["size_lsb"] != tmp_size + j or e["header_idx"] != tmp_head_idx):
                    break
                j += 1
                i += 1
            if j != tmp_mul - tmp_size:
                tmp_fields = 6
            b += put_v(tmp_flags) + put_v(tmp_fields)
            if tmp_fields > 0: b += put_s(tmp_pts)
            if tmp_fields > 1: b += put_v(tmp_mul)
            if tmp_fields > 2: b += put_v(tmp_stream)
            if tmp_fields > 3: b += put_v(tmp_size)
            if tmp_fields > 4: b += put_v(0)
            if tmp_fields > 5: b += put_v(j)
            if tmp_fields > 6: b += put_v(tmp_match)
            if tmp_fields > 7: b += put_v(tmp_head_idx)
        b += put_v(self.header_count - 1)
        for hdr in (b"\x00\x00\x01", b"\x00\x00\x01\xB6", b"\xFF\xFA", b"\xFF\xFB", b"\xFF\xFC", b"\xFF\xFD"):
            b += put_v(len(hdr)) + hdr
        return bytes(b)

    def _streamheader(self):
        b = bytearray()
        b += put_v(0) + put_v(0)                     # stream id, class video
        b += put_v(4) + self.tag.to_bytes(4, "little")
        b += put_v(0)                                # time base index
        b += put_v(self.msb_pts_shift) + put_v(self.max_pts_distance)
        b += put_v(0)                                # video_delay
        b += bytes([0])                              # flags
        b += put_v(0)                                # extradata size
        b += put_v(self.w) + put_v(self.h)
        if self.sar[0] <= 0 or self.sar[1] <= 0:
            b += put_v(0) + put_v(0)
        else:
            b += put_v(self.sar[0]) + put_v(self.sar[1])
        b += put_v(0)                                # csp type
        return bytes(b)

    def _write_headers(self):
        o = self.out
        o += put_packet(self._mainheader(), MAIN_STARTCODE)
        o += put_packet(self._streamheader(), STREAM_STARTCODE)
        o += put_packet(put_v(0) + put_v(0) + put_v(0) + put_v(0) + put_v(0), INFO_STARTCODE)      # global info: no metadata with -fflags +bitexact
        items = [(k, v) for k, v in self.meta] + [("r_frame_rate", f"{self.frame_rate[0]}/{self.frame_rate[1]}")]
        info = bytearray(put_v(1) + put_v(0) + put_v(0) + put_v(0) + put_v(len(items)))
        for k, v in items:
            info += put_str(k) + put_s(-1) + put_str(v)
        o += put_packet(bytes(info), INFO_STARTCODE)
        self.last_syncpoint_pos = -(1 << 31)
        self.header_count += 1

    def write_header(self):
        self.out += ID_STRING                          # (ID_STRING's own terminator + avio_w8(0))
        self._write_headers()

    # ---- packets ----
    def _needed_flags(self, fc, size, pts, key):
        flags = 0
        if key: flags |= FLAG_KEY
        if fc["stream_id"] != 0: flags |= FLAG_STREAM_ID
        if size // fc["size_mul"]: flags |= FLAG_SIZE_MSB
        if pts - self.last_pts != fc["pts_delta"]: flags |= FLAG_CODED_PTS
        if size > 2 * MAX_DISTANCE: flags |= FLAG_CHECKSUM
        if abs(pts - self.last_pts) > self.max_pts_distance: flags |= FLAG_CHECKSUM
        # (header_idx is 0 for every video frame code)
        return flags | (fc["flags"] & FLAG_CODED)

    def write_packet(self, data, pts_in, key=True):
        pts = pts_in * self.in_tb[0] * self.tb[1] // (self.in_tb[1] * self.tb[0])     # av_packet_rescale_ts to the stream time base (exact here)
        dts = pts
        o = self.out
        size = len(data)
        if (1 << (20 + 3 * self.header_count)) <= len(o):
            self._write_headers()
        store_sp = bool(key and not (self.last_flags & FLAG_KEY))
        if size + 30 + len(o) >= self.last_syncpoint_pos + MAX_DISTANCE:
            store_sp = True
        if store_sp:
            self.last_pts = dts                        # ff_nut_reset_ts (one time base)
            sp_pos = None
            cand = [pos for pos, p in self.index_entries if p <= dts]      # av_index_search_timestamp(..., AVSEEK_FLAG_BACKWARD)
            if cand:
                sp_pos = cand[-1]
            self.last_syncpoint_pos = len(o)
            payload = put_v(dts * 1 + 0) + put_v(((self.last_syncpoint_pos - sp_pos) >> 4) if sp_pos is not None else 0)
            o += put_packet(payload, SYNCPOINT_STARTCODE)
            self.sp.append(self.last_syncpoint_pos)
            n = len(self.sp)
            if (1 << 60) % n == 0:                     # sp_count a power of two: the keyframe table doubles
                start = 0 if n == 1 else n
                self.keyframe_pts += [NOPTS] * (2 * n - len(self.keyframe_pts))
                for j in range(start, 2 * n):
                    self.keyframe_pts[j] = NOPTS
        coded_pts = pts & ((1 << self.msb_pts_shift) - 1)
        # ff_lsb2full: the full value nearest to last_pts with these low bits
        mask = (1 << self.msb_pts_shift) - 1
        delta = self.last_pts - mask // 2
        full = ((coded_pts - delta) & mask) + delta
        if full != pts:
            coded_pts = pts + (1 << self.msb_pts_shift)
        best_length, frame_code = 1 << 31, -1
        for i in range(256):
            fc = self.fc[i]
            flags = fc["flags"]
            if flags & FLAG_INVALID:
                continue
            needed = self._needed_flags(fc, size, pts, key)
            length = 0
            if flags & FLAG_CODED:
                length += 1
                flags = needed
            if (flags & needed) != needed:
                continue
            if (flags ^ needed) & FLAG_KEY:
                continue
            if flags & FLAG_STREAM_ID:
                length += v_length(0)
            if size % fc["size_mul"] != fc["size_lsb"]:
                continue
            if flags & FLAG_SIZE_MSB:
                length += v_length(size // fc["size_mul"])
            if flags & FLAG_CHECKSUM:
                length += 4
            if flags & FLAG_CODED_PTS:
                length += v_length(coded_pts)
            # (best_header_idx is 0 for raw video: no elision header matches, header_len[0] = 0)
            if flags & FLAG_HEADER_IDX:
                length += 1
            length *= 4
            length += 0 if flags & FLAG_CODED_PTS else 1
            length += 0 if flags & FLAG_CHECKSUM else 1
            if length < best_length:
                best_length, frame_code = length, i
        assert frame_code != -1
        fc = self.fc[frame_code]
        flags = fc["flags"]
        needed = self._needed_flags(fc, size, pts, key)
        head = bytearray([frame_code])
        if flags & FLAG_CODED:
            head += put_v((flags ^ needed) & ~FLAG_CODED)
            flags = needed
        if flags & FLAG_STREAM_ID: head += put_v(0)
        if flags & FLAG_CODED_PTS: head += put_v(coded_pts)
        if flags & FLAG_SIZE_MSB: head += put_v(size // fc["size_mul"])
        if flags & FLAG_HEADER_IDX: head += put_v(0)
        if flags & FLAG_CHECKSUM:
            head += crc32_ieee_msb(bytes(head)).to_bytes(4, "big")
        o += head
        o += data
        self.last_flags = flags
        self.last_pts = pts
        if flags & FLAG_KEY:
            self.index_entries.append((self.last_syncpoint_pos, pts))
            n = len(self.sp)
            if self.keyframe_pts and n < len(self.keyframe_pts) and self.keyframe_pts[n] == NOPTS:
                self.keyframe_pts[n] = pts
        if self.max_pts is None or self.max_pts < pts:
            self.max_pts = pts

    # ---- trailer ----
    def write_trailer(self):
        while self.header_count < 3:
            self._write_headers()
        n = len(self.sp)
        if not n:
            return
        b = bytearray()
        b += put_v(self.max_pts * 1 + 0)
        b += put_v(n)
        prev = 0
        for pos in self.sp:
            b += put_v((pos >> 4) - (prev >> 4))
            prev = pos
        kf = list(self.keyframe_pts)
        last_pts = -1
        j = 0
        while j < n:
            if j and kf[j] == kf[j - 1]:
                kf[j] = NOPTS
            flag = int(kf[j] != NOPTS) ^ int(j + 1 == n)
            cnt = 0
            while j < n and int(kf[j] != NOPTS) == flag:
                cnt += 1
                j += 1
            b += put_v(1 + 2 * flag + 4 * cnt)
            k = j - cnt
            while k <= j and k < n:
                if kf[k] != NOPTS:
                    assert kf[k] > last_pts
                    b += put_v(kf[k] - last_pts)
                    last_pts = kf[k]
                k += 1
            j += 1
        payload_size = len(b) + 8 + 4
        b += (8 + payload_size + (payload_size.bit_length() - 1) // 7 + 1 + 4 * (1 if payload_size > 4096 else 0)).to_bytes(8, "big")
        self.out += put_packet(bytes(b), INDEX_STARTCODE)


def nut_md5(frames, width, height, codec_tag, **kw):
    """MD5 of the NUT file holding `frames` (bytes of one raw picture each, pts 0, 1, ... in 1/25 s)"""
    m = NutVideoMuxer(width, height, codec_tag, **kw)
    m.write_header()
    for i, f in enumerate(frames):
        m.write_packet(f, i, True)
    m.write_trailer()
    return hashlib.md5(bytes(m.out)).hexdigest(), bytes(m.out)
