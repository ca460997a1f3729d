from .yolov5_neck import YoloV5Neck


def build_neck(cfg):
    # reference models/neck/__init__.py:23
    if cfg.Model.Neck.name == 'YoloV5':
        return YoloV5Neck(cfg)
    if cfg.Model.Neck.name == 'YoloV8':
        from .yolov8_neck import YoloV8Neck
        return YoloV8Neck(cfg)
    raise NotImplementedError(f"neck {cfg.Model.Neck.name}: only the YoloV5 hot path is built")
