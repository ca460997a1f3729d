from .yolov5_neck import YoloV5Neck


def build_neck(cfg):
    # reference models/neck/__init__.py:23
    if cfg.Model.Neck.name == 'YoloV5':
        return YoloV5Neck(cfg)
    raise NotImplementedError(f"neck {cfg.Model.Neck.name}: only the YoloV5 hot path is built")
