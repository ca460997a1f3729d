"""TEST INFRASTRUCTURE -- CPU restatement of the Detect head (anchor-based).

Restates ``Detect.forward`` (models/head/yolov5_head.py:47-87) and
``_make_grid_old`` (:127-136) in plain torch fp32:
  train : x[i] = conv_i(f_i).view(B,na,no,ny,nx).permute(0,1,3,4,2).contiguous()
  eval  : y = sigmoid(x); xy = (2*y-0.5+grid)*stride; wh = (2*y)^2*anchors*stride;
          z = cat_i y.view(B,-1,no)  ->  (z, x)
``anchors`` is the stride-normalised (nl,na,2) buffer (yolo_ssod.py:81).
"""
import torch


def permute_raw(raw, na, no):
    """(B, na*no, ny, nx) -> (B, na, ny, nx, no) contiguous (yolov5_head.py:66)."""
    b, _, ny, nx = raw.shape
    return raw.view(b, na, no, ny, nx).permute(0, 1, 3, 4, 2).contiguous()


def decode(xs, anchors, strides):
    """xs: list of (B,na,ny,nx,no) raw logits -> (B, sum na*ny*nx, no)."""
    z = []
    for i, x in enumerate(xs):
        b, na, ny, nx, no = x.shape
        yv, xv = torch.meshgrid([torch.arange(ny), torch.arange(nx)], indexing="ij")
        grid = torch.stack((xv, yv), 2).expand(1, na, ny, nx, 2).float()
        ag = (anchors[i].clone() * strides[i]).view(1, na, 1, 1, 2).expand(1, na, ny, nx, 2).float()
        y = x.sigmoid()
        y[..., 0:2] = (y[..., 0:2] * 2. - 0.5 + grid) * strides[i]
        y[..., 2:4] = (y[..., 2:4] * 2) ** 2 * ag
        z.append(y.view(b, -1, no))
    return torch.cat(z, 1)
