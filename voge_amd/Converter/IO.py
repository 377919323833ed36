"""OFF / COFF / GOFF text IO (VoGE/Converter/IO.py).

GOFF layout (IO.py:5-8,61-88,129-163): line 1 `GOFF`; line 2 `n_vertices sigma_width has_radian`;
n centre lines; n sigma lines of width 1, 3, 6 or 9; optionally n radian lines."""
import numpy as np
import torch


def _rows(lines, dtype):
    return [np.array(ln.split(), dtype=dtype) for ln in lines if ln.strip()]


def load_off(file_name, to_torch=False, ignore_color=False):
    """-> (verts [n,3] f32, faces [m,3] i32[, vert_color][, face_color])."""
    with open(file_name) as fh:
        lines = fh.readlines()
    tag = lines[0].strip()
    if ignore_color or tag.startswith('OFF'):
        colored = False
    elif tag.startswith('COFF'):
        colored = True
    else:
        raise Exception('Unsupported OFF format: %s' % tag)
    n_pts, n_faces = (int(v) for v in lines[1].split()[:2])
    vrows = np.stack(_rows(lines[2:2 + n_pts], np.float32)) if n_pts else np.zeros((0, 3), np.float32)
    frows = np.stack(_rows(lines[2 + n_pts:2 + n_pts + n_faces], np.float64)) if n_faces else np.zeros((0, 4))
    out = [vrows[:, :3], None]
    if colored and vrows.shape[1] > 3:
        out.append(vrows[:, 3:])
    nv = int(frows[0, 0]) if n_faces else 3
    out[1] = frows[:, 1:nv + 1].astype(np.int32)
    if colored and frows.shape[1] > nv + 1:
        out.append(frows[:, nv + 1:].astype(np.int32))
    return tuple(torch.from_numpy(np.ascontiguousarray(t)) for t in out) if to_torch else tuple(out)


def save_off(file_name, vertices, faces, vert_color=None, face_color=None):
    vertices, faces, vert_color, face_color = (t.cpu().numpy() if isinstance(t, torch.Tensor) else t
                                               for t in (vertices, faces, vert_color, face_color))
    with open(file_name, 'w') as fh:
        fh.write('OFF\n' if vert_color is None and face_color is None else 'COFF\n')
        fh.write('%d %d 0\n' % (len(vertices), len(faces)))
        for i, v in enumerate(vertices):
            vals = list(v[:3]) + ([] if vert_color is None else list(vert_color[i]))
            fh.write(' '.join('%.16f' % x for x in vals) + '\n')
        for i, f in enumerate(faces):
            fh.write('3 %d %d %d' % tuple(f[:3]))
            if face_color is not None:
                fh.write(''.join(' %.16f' % x for x in face_color[i]))
            fh.write('\n')


def load_goff(file_name, to_torch=False):
    with open(file_name) as fh:
        lines = fh.readlines()
    n, width, has_rad = (int(v) for v in lines[1].split()[:3])
    pts = np.stack(_rows(lines[2:2 + n], np.float32)).reshape(-1, 3)
    sig = np.stack(_rows(lines[2 + n:2 + 2 * n], np.float32)).reshape(-1, width)
    if width == 9:
        sig = sig.reshape(-1, 3, 3)
    rad = np.concatenate(_rows(lines[2 + 2 * n:], np.float32)) if has_rad else None
    if to_torch:
        return torch.from_numpy(pts), torch.from_numpy(sig), None if rad is None else torch.from_numpy(rad)
    return pts, sig, rad


def save_goff(file_name, points, sigmas, radians=None):
    points, sigmas, radians = (t.cpu().numpy() if isinstance(t, torch.Tensor) else t for t in (points, sigmas, radians))
    sigmas = np.asarray(sigmas)
    sigmas = sigmas.reshape(len(sigmas), -1)
    with open(file_name, 'w') as fh:
        fh.write('GOFF\n%d %d %d\n' % (len(points), sigmas.shape[1], 0 if radians is None else 1))
        for arr in (points, sigmas):
            for row in arr:
                fh.write(' '.join('%.16f' % x for x in np.ravel(row)) + '\n')
        if radians is not None:
            for x in radians:
                fh.write('%.16f\n' % x)


def to_torch(*args):
    return [torch.from_numpy(t).type(torch.float32) if t is not None else None for t in args]
