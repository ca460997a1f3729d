from .yolov5_head import Detect


def build_head(cfg):
    # reference models/head/__init__.py:12
    if cfg.Model.Head.name == 'YoloV5':
        return Detect(cfg)
    if cfg.Model.Head.name == 'YoloV8':
        from .yolov8_head import YoloV8Detect
        return YoloV8Detect(cfg)
    raise NotImplementedError(f"head {cfg.Model.Head.name}: only the YoloV5 hot path is built")
