"""Generated CW tokens -> Standard MIDI File.

`write_midi(words, path_outfile, word2event)` has the signature and the token semantics of the reference's
writer (dqn_policy/testing-no-type-cp.py:56-123, dqn_policy/agent_pretrain.py:66-129): a token whose pitch,
duration and velocity events are all strings is a note at the current position; otherwise it is a metrical
token -- 'Bar' advances the bar counter, 'Beat_<k>' sets the position to bar*1920 + k*120 ticks and may carry a
chord marker and a tempo change.  The reference serialises through miditoolkit (absent from this image, and a
pure file-format dependency); here the same events are written directly as a format-1 SMF: track 0 = tempo
changes + chord markers, track 1 = the piano notes.  `read_smf` parses what `write_smf` writes (tests).
Host-side only; nothing here touches the GPU.
"""
import struct

BEAT_RESOL = 480
BAR_RESOL = BEAT_RESOL * 4
TICK_RESOL = BEAT_RESOL // 4


def words_to_events(words, word2event):
    """-> dict(notes=[(pitch, start, end, velocity)], markers=[(time, text)], tempo_changes=[(time, bpm)])."""
    keys = list(word2event.keys())
    bar_cnt, cur_pos = 0, 0
    notes, markers, tempos = [], [], []
    for w in words:
        vals = [word2event[k][int(w[i])] for i, k in enumerate(keys)]
        is_note = isinstance(vals[3], str) and isinstance(vals[4], str) and isinstance(vals[5], str)
        if not is_note:
            if vals[2] == "Bar":
                bar_cnt += 1
            if vals[2] == 0:
                pass
            elif "Beat" in vals[2]:
                cur_pos = bar_cnt * BAR_RESOL + int(vals[2].split("_")[1]) * TICK_RESOL
                if vals[1] != "CONTI" and vals[1] != 0:
                    markers.append((cur_pos, str(vals[1])))
                if vals[0] != "CONTI" and vals[0] != 0:
                    tempos.append((cur_pos, int(vals[0].split("_")[-1])))
        else:
            try:
                pitch, dur, vel = (int(v.split("_")[-1]) for v in vals[3:6])
            except ValueError:
                continue
            if dur == 0:
                dur = 60
            notes.append((pitch, cur_pos, cur_pos + dur, vel))
    return {"notes": notes, "markers": markers, "tempo_changes": tempos}


def _vlq(n):
    out = [n & 0x7F]
    n >>= 7
    while n:
        out.append((n & 0x7F) | 0x80)
        n >>= 7
    return bytes(reversed(out))


def _track(events):
    """events: [(tick, order, bytes)] -> MTrk chunk (delta times, end-of-track appended)."""
    body, last = bytearray(), 0
    for tick, _, data in sorted(events, key=lambda e: (e[0], e[1])):
        body += _vlq(tick - last) + data
        last = tick
    body += b"\x00\xff\x2f\x00"
    return b"MTrk" + struct.pack(">I", len(body)) + bytes(body)


def write_smf(path, notes, markers=(), tempo_changes=(), ticks_per_beat=BEAT_RESOL, track_name="piano"):
    meta = []
    for t, bpm in tempo_changes:
        us = int(round(60000000.0 / max(int(bpm), 1)))
        meta.append((t, 0, b"\xff\x51\x03" + struct.pack(">I", us)[1:]))
    for t, text in markers:
        raw = text.encode("utf-8")
        meta.append((t, 1, b"\xff\x06" + _vlq(len(raw)) + raw))
    name = track_name.encode("utf-8")
    ev = [(0, 0, b"\xff\x03" + _vlq(len(name)) + name), (0, 1, b"\xc0\x00")]
    for pitch, start, end, vel in notes:
        p, v = min(max(int(pitch), 0), 127), min(max(int(vel), 1), 127)
        ev.append((int(start), 3, bytes([0x90, p, v])))
        ev.append((int(max(end, start)), 2, bytes([0x80, p, 0])))       # offs sort before ons of the same tick
    with open(path, "wb") as f:
        f.write(b"MThd" + struct.pack(">IHHH", 6, 1, 2, ticks_per_beat) + _track(meta) + _track(ev))


def write_midi(words, path_outfile, word2event):
    e = words_to_events(words, word2event)
    write_smf(path_outfile, e["notes"], e["markers"], e["tempo_changes"])
    return e


def read_smf(path):
    """Parse a file written by write_smf -> dict(ticks_per_beat, notes, markers, tempo_changes)."""
    data = open(path, "rb").read()
    if data[:4] != b"MThd":
        raise ValueError("not a Standard MIDI File")
    _, fmt, ntrk, tpb = struct.unpack(">IHHH", data[4:14])
    pos = 14
    notes, markers, tempos, open_notes = [], [], [], {}
    for _ in range(ntrk):
        if data[pos:pos + 4] != b"MTrk":
            raise ValueError("bad track chunk")
        n = struct.unpack(">I", data[pos + 4:pos + 8])[0]
        p, end, tick = pos + 8, pos + 8 + n, 0
        while p < end:
            d = 0
            while True:
                b = data[p]
                p += 1
                d = (d << 7) | (b & 0x7F)
                if not b & 0x80:
                    break
            tick += d
            st = data[p]
            if st == 0xFF:
                kind, ln, q = data[p + 1], 0, p + 2
                while True:
                    b = data[q]
                    q += 1
                    ln = (ln << 7) | (b & 0x7F)
                    if not b & 0x80:
                        break
                payload = data[q:q + ln]
                if kind == 0x51:
                    tempos.append((tick, int(round(60000000.0 / int.from_bytes(payload, "big")))))
                elif kind == 0x06:
                    markers.append((tick, payload.decode("utf-8")))
                p = q + ln
            elif st & 0xF0 == 0xC0:
                p += 2
            elif st & 0xF0 == 0x90:
                open_notes.setdefault(data[p + 1], []).append((tick, data[p + 2]))
                p += 3
            elif st & 0xF0 == 0x80:
                s, v = open_notes[data[p + 1]].pop(0)
                notes.append((data[p + 1], s, tick, v))
                p += 3
            else:
                raise ValueError("unexpected status byte 0x%02x" % st)
        pos = end
    return {"ticks_per_beat": tpb, "notes": sorted(notes, key=lambda x: (x[1], x[0])), "markers": markers,
            "tempo_changes": tempos}
