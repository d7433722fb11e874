"""Drop-in import surface: ``from model.deformable_detr import ...`` / ``from model.egtr import ...`` exactly as
the reference's train_egtr.py:31-36 and evaluate_egtr.py do, resolved to the MI355X-native package egtr_amd."""
