"""Minimal model registry with the reference's access pattern
(``registry.get_model_class("st_llm_hf").from_config(cfg)``, common/registry.py:83-110, demo.py:44)."""


class Registry:
    mapping = {"model_name_mapping": {}}

    @classmethod
    def register_model(cls, name):
        def wrap(model_cls):
            if name in cls.mapping["model_name_mapping"]:
                raise KeyError(f"Name '{name}' already registered")
            cls.mapping["model_name_mapping"][name] = model_cls
            return model_cls
        return wrap

    @classmethod
    def get_model_class(cls, name):
        return cls.mapping["model_name_mapping"].get(name, None)


registry = Registry()
