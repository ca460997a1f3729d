from .yolov5_backbone import YoloV5BackBone


def build_backbone(cfg):
    # reference models/backbone/__init__.py:8 dispatches on cfg.Model.Backbone.name
    if cfg.Model.Backbone.name == 'YoloV5':
        return YoloV5BackBone(cfg)
    if cfg.Model.Backbone.name == 'YoloV8':
        from .yolov8_backbone import YoloV8BackBone
        return YoloV8BackBone(cfg)
    raise NotImplementedError(f"backbone {cfg.Model.Backbone.name}: only the YoloV5 hot path is built "
                              f"(SURVEY.md section 8 scope)")
